#!/bin/bash
# GPU session H of round 5: C3 -- the normaliser's per-lane totals (per-byte counts only where a lane is not plain) and the wavefront's own
# candidate list in the end-mask lookup -- against the build before (tools/ab_libs/r5_prev.so); the BERT-side hardware tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p "$O"
timeout 600 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_prev.so" "" "AB_LIB=tools/ab_libs/r5_prev.so" 2>&1 | tee "$O/ab_c3.txt"
timeout 900 python -m pytest tests -m gpu -q -n 4 -k "bert or wordpiece or wordlevel or bpe_over or claims or norm or c3 or epilogue" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
timeout 300 python tools/char_bpe_perf.py > "$O/char_bpe_perf.txt" 2>&1; tail -4 "$O/char_bpe_perf.txt"
