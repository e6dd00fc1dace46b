#!/bin/bash
# GPU session P of round 4: C3 without the 360 MB memset of the normalised text's buffer (only the slack behind the text is zeroed), the
# four match masks zeroed by one launch (and only as far as the raw text reaches when that is what they cover).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py tests/test_epilogue_gpu.py tests/test_pretokenized_gpu.py -m gpu -q -x -k "added or c3_bert or golden or wordpiece or special or quirk or random or bert or normal" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" 2>&1 | tee "$O/ab_c3.txt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_c3" -- python bench.py --config c3 --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 10 --warmup 2 > "$O/stats_c3.log" 2>&1; S=$(ls $O/stats_c3/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/c3_kernel_stats.csv"; rm -rf "$O/stats_c3"
timeout 300 python tools/char_bpe_perf.py bpe_bert_affixes 2>&1 | tail -2 | tee "$O/char_bpe_bert.txt"
