#!/usr/bin/env python3
"""tg_step_many_i8 at the config sizes (hipGraph replay, events): python tools/many_bench.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for (s2, b2, k2, vals, probs) in [(25, 4096, 64, (-1, 0, 1), (0.15, 0.7, 0.15)), (16, 8192, 20, (-1, 0, 1), (0.15, 0.7, 0.15)),
                                   (16, 8192, 64, (-1, 0, 1), (0.15, 0.7, 0.15)),
                                   (25, 4096, 64, (-2, -1, 0, 1, 2), (0.05, 0.1, 0.7, 0.1, 0.05)), (25, 4096, 200, (-1, 0, 1), (0.15, 0.7, 0.15))]:
    tok, tgt = ops.gen_demos(b2, s2, k2, dev, values=vals, probs=probs, seed=2)
    st2 = ops.alloc_states(b2, s2, dev)
    ds = torch.zeros(b2, dtype=torch.int32, device=dev)
    h0 = ops.debug_handovers(dev)
    sec = bench.graph_time(lambda: ops.step_many(tgt, tok, out=st2, done_step=ds), dev, reps=20)
    print(f"S={s2} B={b2} K={k2} values={vals}: {sec * 1e6:.2f} us  zero={not bool(st2.any())}  handovers per launch={(ops.debug_handovers(dev) - h0) / (3 + 20 * 7):.1f}")
