#!/usr/bin/env python3
"""step_many: time per launch over (S, K) -- run once with and once without TG_NO_MFMA=1 to place the
dispatch threshold between the matrix-core path and the lattice kernels."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from mat_mul_amd import ops

dev = "cuda:0"
for S, B in ((9, 32768), (16, 8192), (25, 4096)):
    for K in (4, 8, 16, 20, 32, 64, 128, 256):
        tok, tgt = ops.gen_demos(B, S, K, dev, seed=1)
        out = ops.alloc_states(B, S, dev)
        ds = torch.zeros(B, dtype=torch.int32, device=dev)
        for _ in range(3):
            ops.step_many(tgt, tok, out=out, done_step=ds)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.step_many(tgt, tok, out=out, done_step=ds)
        e1.record()
        torch.cuda.synchronize()
        print(S, B, K, round(e0.elapsed_time(e1) * 100, 1), "us", "flagged/early:", int((ds != K - 1).sum()), flush=True)
