#!/usr/bin/env python3
"""tg_step_many_i8 over K (graph of launches, events): where does the matrix-core pass start to pay?
    [TG_LIB_VARIANT=ab TG_MFMA_MANY_ALWAYS=1] python tools/many_k_sweep.py"""
import sys; sys.path.insert(0, '/root/repo')
import torch
import bench
from mat_mul_amd import ops
dev = torch.device("cuda", 0)
for (s2, b2, k2) in [(16, 8192, 8), (16, 8192, 12), (16, 8192, 16), (16, 8192, 20), (16, 8192, 24), (16, 8192, 32), (16, 8192, 40), (25, 4096, 2), (25, 4096, 3), (25, 4096, 4), (25, 4096, 8), (25, 4096, 20), (9, 32768, 12), (9, 32768, 24), (9, 32768, 32), (9, 32768, 48)]:
    tok, tgt = ops.gen_demos(b2, s2, k2, dev, seed=2)
    st2 = ops.alloc_states(b2, s2, dev)
    ds = torch.zeros(b2, dtype=torch.int32, device=dev)
    sec = bench.graph_time(lambda: ops.step_many(tgt, tok, out=st2, done_step=ds), dev, reps=10)
    print(f"S={s2} B={b2} K={k2}: {sec * 1e6:.2f} us  zero={not bool(st2.any())}")
