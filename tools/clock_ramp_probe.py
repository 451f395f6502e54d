#!/usr/bin/env python3
"""Does the shader clock ramp during a sustained run?  Launches tg_step_many_i8 (S=25, B=4096, K=64: VALU / matrix-core
bound) back to back for ~1 s and prints the per-launch time (events) of successive windows, then the same for the fused
generator.  python tools/clock_ramp_probe.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
S, B, K = 25, 4096, 64
tok, tgt = ops.gen_demos(B, S, K, dev, seed=2)
out = ops.alloc_states(B, S, dev)
ds = torch.zeros(B, dtype=torch.int32, device=dev)
ovf = torch.zeros(B, dtype=torch.uint8, device=dev)


def windows(fn, label, nwin=12, per=400):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nwin + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for w in range(nwin):
        for _ in range(per):
            fn()
        ev[w + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 / per for i in range(nwin)]
    print(f"{label}: per-launch us by window of {per} launches: " + " ".join(f"{u:.1f}" for u in us) + f"   (wall {wall * 1e3:.0f} ms)")


windows(lambda: ops.step_many(tgt, tok, out=out, done_step=ds, overflow=ovf), "step_many S=25 K=64 B=4096")
windows(lambda: ops.gen_demos(B, S, K, dev, seed=2, target=tgt, actions=tok, overflow=ovf), "gen_demos S=25 R=64 B=4096")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50):
        ops.step_many(tgt, tok, out=out, done_step=ds, overflow=ovf)
windows(g.replay, "step_many, graph of 50", nwin=12, per=8)
