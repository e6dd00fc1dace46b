#!/usr/bin/env python3
"""Where does k_claims_mark spend its time?  One C2 batch, the kernel with parts switched off (TKAMD_CM_DEBUG bits: 1 no claim reads,
2 no CAS, 4 no compare of the claimant's bytes, 8 no rank loads / tok0 stores, 16 no dead-entry stores, 32 no second slot; results are
WRONG with any of them -- timing only) and with 1 / 2 / 4 entries per lane (TKAMD_CM_K).   usage: python tools/cm_probe.py [c2] [type_seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import tokenizers_amd as ta

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
ts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
js, n_types, _ = bench.load_config(cfg)
tok = ta.Tokenizer.from_str(js, device=0)
dev = torch.device("cuda", 0)
b = bench.Batch(bench.make_corpus(cfg, 1_000_000, 100, ts, n_types), dev, 0, False)
stream = torch.cuda.current_stream().cuda_stream
enc = lambda: tok.encode_batch_device(b.d_text.data_ptr(), b.d_off.data_ptr(), b.n_docs, b.n_bytes, stream=stream)
for k, dbg in [(1, 0), (1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (1, 24), (1, 32), (1, 60), (1, 63), (2, 0), (4, 0), (4, 24), (4, 60), (4, 63)]:
    os.environ["TKAMD_CM_K"], os.environ["TKAMD_CM_DEBUG"] = str(k), str(dbg)
    for _ in range(2):
        enc()
    enc().sync()
    tok.profile(True)
    for _ in range(8):
        enc()
    enc().sync()
    tok.profile(False)
    st = {kk: round(v[0] / max(1, v[1]), 4) for kk, v in tok.profile_read().items()}
    print(f"{cfg} ts={ts} K={k} debug={dbg:2d}: claims_mark {st.get('claims_mark')} ms  claims_compact {st.get('claims_compact')}  merges {st.get('bpe_merge_lds32')} {st.get('bpe_merge_lds')}  queues {tok.queue_sizes()['merge16']}", flush=True)
