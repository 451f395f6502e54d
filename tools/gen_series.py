#!/usr/bin/env python3
"""The fused generator at BASELINE config 5's per-GPU share, timed again and again in one process (hipGraph of 20
launches, events): how long does the GPU take to reach its steady rate?   python tools/gen_series.py [basis] [n]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
S, B, R = 25, 4096, 64
basis = "basis" in sys.argv[1:]
n = next((int(a) for a in sys.argv[1:] if a.isdigit()), 40)
P = ops.sample_basis(B, S, dev, seed=3) if basis else None
tok = torch.empty((B, R, 3 * S), dtype=torch.int8, device=dev)
tgt = ops.alloc_states(B, S, dev)
ovf = torch.zeros(B, dtype=torch.uint8, device=dev)
fn = lambda: ops.gen_demos(B, S, R, dev, seed=1, basis=P, target=tgt, actions=tok, overflow=ovf)  # noqa: E731
t0 = time.perf_counter()
out = []
for i in range(n):
    out.append((time.perf_counter() - t0, bench.graph_time(fn, dev, reps=20) * 1e6))
print(" ".join(f"{t * 1e3:.0f}ms:{u:.2f}" for t, u in out))
# one long graph replayed back to back: per-replay times
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.graph(g, stream=side):
    for _ in range(100):
        fn()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize(dev)
time.sleep(0.5)   # let the GPU idle
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
for i in range(41):
    evs[i].record()
    if i < 40:
        g.replay()
torch.cuda.synchronize(dev)
print("after 0.5 s idle, replays of 100 launches (us per launch):", " ".join(f"{evs[i].elapsed_time(evs[i + 1]) * 10:.2f}" for i in range(40)))
