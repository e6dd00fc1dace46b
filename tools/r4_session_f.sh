#!/bin/bash
# GPU session F of round 4: the whole hardware gate on the current tree (BPE over characters included), then what that new path costs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4f; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$O/pytest_gpu.txt"
for n in bpe_ws_unk bpe_bert_affixes bpe_ws_byte_fallback; do timeout 200 python tools/char_bpe_perf.py $n 2>&1 | tail -2; done | tee "$O/char_bpe_perf.txt"
