#!/bin/bash
# One GPU-box session that produces everything a round commits under profiles/ (copy gpurun_out/<tag>/* there afterwards, as <tag>_*):
# the hardware gate, the driver-style bench line (C2 with every leg; C3..C5 ride in its `other_configs`), rocprofv3 --kernel-trace --stats
# per config, smoke(), and -- last, a bench line right behind a --pmc pass comes out slow -- the FETCH / WRITE passes per config and the
# SQ / TCC counters of C2's dominant kernels (ids-only and with offsets).
# usage (on the GPU box, from the repo root): tools/round_profile.sh <tag> [skip-gate]        e.g. r6
tag=${1:-r6}
gate=gate; [ "$2" = "skip-gate" ] && gate=""
exec "$(dirname "$0")/session.sh" "$tag" $gate bench stats:c2 stats:c3 stats:c4 stats:c5 smoke pmc:c2 pmc:c3 pmc:c4 pmc:c5 \
     sq:c2 "sq:c2:k_token_meta,k_emit_pretok,k_leadmask,k_compact:byte"
