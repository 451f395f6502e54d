#!/usr/bin/env python3
"""Diagnostic build only (libtensorgame_stamps.so): where does one workgroup of many_mfma_kernel (tg_step_many_i8 on
the matrix cores) spend its cycles?  Workgroups 0, 64, .., 960 record s_memtime (2.4 GHz) at: 0 entry (s_memrealtime), 1 set-up done, then per game:
staged, B1 passed, scalars done, barrier + bound, tiles done, B2 passed, verdict + stores issued; last: after the final
barrier; 31: exit (s_memrealtime, 100 MHz).  Prints the median over the workgroups of each interval in shader cycles.
Run:  TG_LIB_VARIANT=stamps python tools/stamp_many.py [wide]"""
import os
import sys
from pathlib import Path

os.environ["TG_LIB_VARIANT"] = "stamps"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402

S, B, R = 25, 4096, 64
dev = "cuda:0"
wide = len(sys.argv) > 1 and sys.argv[1] == "wide"
vals, probs = ((-2, -1, 0, 1, 2), (0.05, 0.1, 0.7, 0.1, 0.05)) if wide else ((-1, 0, 1), (0.15, 0.7, 0.15))
ovf = torch.zeros(B, dtype=torch.uint8, device=dev)  # (the stamps build of the generator writes its stamps here too)
tok, tgt = ops.gen_demos(B, S, R, dev, values=vals, probs=probs, seed=2, overflow=ovf)
out = ops.alloc_states(B, S, dev)
ds = torch.zeros(B, dtype=torch.int32, device=dev)
ovf.zero_()
for _ in range(3):
    ops.step_many(tgt, tok, out=out, done_step=ds, overflow=ovf)
torch.cuda.synchronize()
st = ovf.view(torch.int64)[:512].reshape(16, 32).cpu()
phases = ["stage (tokens, state -> LDS)", "wait B1", "scalars", "barrier + bound", "tiles", "wait B2", "verdict + stores"]
names = ["entry->setup"] + [f"g{g}: {p}" for g in range(4) for p in phases] + ["final barrier"]
t1 = st[:, 31].min()
print("s_memrealtime (100 MHz) at exit, relative to the first workgroup to finish, us:", [round(float(x - t1) / 100, 1) for x in st[:, 31]])
d = st[:, 1:] - st[:, :-1]
for i, n in enumerate(names):
    col = d[:, i]
    print(f"{n:40s} median {int(col.median()):7d}  min {int(col.min()):7d}  max {int(col.max()):7d}")
print("total entry -> last stamp: median", int((st[:, len(names)] - st[:, 0]).median()), "ticks")
print("per phase, summed over the four games (median workgroup):",
      {p: int(sum(d[:, 1 + 7 * g + j].median() for g in range(4))) for j, p in enumerate(phases)})
