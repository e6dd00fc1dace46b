#!/bin/bash
# GPU session N of round 4 (after the profile; no product change since): the legs that are not in the bench line, on HEAD -- the claims'
# worst case, the host entry per slice size, the link, BPE over characters, and the random differential against the wheel ON the device.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n; mkdir -p "$O"
timeout 300 python tools/claims_worst_case.py > "$O/claims_worst_case.txt" 2>&1; tail -4 "$O/claims_worst_case.txt" | cut -c1-200
TKAMD_CLAIMS=0 timeout 300 python tools/claims_worst_case.py > "$O/claims_worst_case_off.txt" 2>&1; tail -2 "$O/claims_worst_case_off.txt" | cut -c1-200
timeout 200 python tools/link_probe.py 128 > "$O/link_probe.txt" 2>&1; cat "$O/link_probe.txt"
timeout 400 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_PHASES=1" "TKAMD_HOT_SLOTS=2048" "TKAMD_PHASES=1 TKAMD_HOT_SLOTS=2048" > "$O/ab_c2.txt" 2>&1; cut -c1-200 "$O/ab_c2.txt"
timeout 400 python tools/host_leg.py 8 16 32 > "$O/host_leg.txt" 2>&1; cat "$O/host_leg.txt"
for n in bpe_ws_unk bpe_bert_affixes bpe_ws_byte_fallback; do timeout 200 python tools/char_bpe_perf.py $n 2>&1 | tail -2; done > "$O/char_bpe_perf.txt"; cut -c1-200 "$O/char_bpe_perf.txt"
TKAMD_FUZZ_GPU=1 timeout 400 python tools/fuzz_live.py 977 300 > "$O/fuzz_gpu.txt" 2>&1; tail -3 "$O/fuzz_gpu.txt"
