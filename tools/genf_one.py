#!/usr/bin/env python3
"""One timing of tg_gen_from_factors_i8 (graph of launches, events): python tools/genf_one.py [S B R]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
S, B, R = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (25, 4096, 64)
tok, _ = ops.gen_demos(B, S, R, dev, seed=1)
out = ops.alloc_states(B, S, dev)
sec = bench.graph_time(lambda: ops.gen_from_factors(tok, S, out=out), dev, reps=20)
print(f"gen_from_factors S={S} B={B} R={R}: {sec * 1e6:.2f} us")
