#!/usr/bin/env python3
"""One timing of the fused generator at BASELINE config 5's per-GPU share (graph of launches, events):
    [TG_LIB_VARIANT=ab TG_GF_WGS=n] python tools/gen_one.py [basis]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
S, B, R = 25, 4096, 64
P = ops.sample_basis(B, S, dev, seed=3) if len(sys.argv) > 1 else None
tok = torch.empty((B, R, 3 * S), dtype=torch.int8, device=dev)
tgt = ops.alloc_states(B, S, dev)
ovf = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = list(bench.graph_time(lambda: ops.gen_demos(B, S, R, dev, seed=1, basis=P, target=tgt, actions=tok, overflow=ovf), dev, reps=20)
            for _ in range(9))
print("in order:", " ".join(f"{t * 1e6:.2f}" for t in ts))
ts = sorted(ts)
print(f"{ts[4] * 1e6:.2f} us (median of 9 timings of 20 launches; min {ts[0] * 1e6:.2f}, max {ts[-1] * 1e6:.2f})")
