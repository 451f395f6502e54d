#!/usr/bin/env python3
"""A/B library only: where does many_mfma_kernel (tg_step_many_i8 on the matrix cores) spend its time?  Runs the kernel
with parts switched off (TG_MANY_ABLATE bit mask: 1 no token staging, 2 no state load, 4 no action scalars, 8 no tiles,
16 no verdict scan, 32 no stores; results are then wrong -- timing only).  Each setting in its own process.
    python tools/ablate_many.py"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys; sys.path.insert(0, sys.argv[1])
import torch, bench
from mat_mul_amd import ops
dev = torch.device("cuda", 0)
S, B, K = 25, 4096, int(sys.argv[2])
tok, tgt = ops.gen_demos(B, S, K, dev, seed=2)
out = ops.alloc_states(B, S, dev); ds = torch.zeros(B, dtype=torch.int32, device=dev)
sec = bench.graph_time(lambda: ops.step_many(tgt, tok, out=out, done_step=ds), dev, reps=10)
print(f"{sec * 1e6:.2f}")
'''
for K in (64, 8):
    print(f"S=25 B=4096 K={K}")
    for bits, what in [(0, "everything"), (1, "no token staging"), (2, "no state load"), (4, "no action scalars"), (8, "no tiles"),
                       (16, "no verdict scan"), (32, "no stores"), (8 + 4, "no tiles, no scalars"), (8 + 4 + 16, "no tiles, scalars, verdict"),
                       (1 + 4 + 8 + 16, "state in, state out only"), (63, "nothing (loop, barriers, set-up)")]:
        env = dict(os.environ, TG_LIB_VARIANT="ab", TG_MANY_ABLATE=str(bits))
        r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), str(K)], env=env, capture_output=True, text=True)
        print(f"  ablate {bits:2d} ({what:34s}): {r.stdout.strip() or r.stderr[-300:]} us", flush=True)
