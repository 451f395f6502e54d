#!/usr/bin/env python3
"""Diagnostic build only (libtensorgame_stamps.so): where does one workgroup of the fused generator spend its cycles?
Workgroups 0..19 record s_memtime at: 0 entry, 1 set-up done, then per game: loop top (stores of the previous game
issued), draw done, B1 passed, tiles done; finally the loop top after the last game.  Prints the median over the
workgroups of each interval, in shader cycles.  Run:  TG_LIB_VARIANT=stamps python tools/stamp_genfused.py [basis]"""
import os
import sys
from pathlib import Path

os.environ["TG_LIB_VARIANT"] = "stamps"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from mat_mul_amd import build, ops  # noqa: E402

S, B, R = 25, 4096, 64
dev = "cuda:0"
basis = len(sys.argv) > 1 and sys.argv[1] == "basis"
P = ops.sample_basis(B, S, dev, seed=3) if basis else None
tok = torch.empty((B, R, 3 * S), dtype=torch.int8, device=dev)
tgt = ops.alloc_states(B, S, dev)
ovf = torch.zeros(B, dtype=torch.uint8, device=dev)
for _ in range(3):
    ops.gen_demos(B, S, R, dev, seed=1, basis=P, target=tgt, actions=tok, overflow=ovf)
torch.cuda.synchronize()
st = ovf.view(torch.int64)[:480].reshape(20, 24).cpu()
names = ["entry->setup"]
for g in range(4):
    names += [f"g{g}: (stores of prev) -> loop top", f"g{g}: draw", f"g{g}: wait B1", f"g{g}: tiles"]
names += ["g3 tiles done -> B2 + final stores issued"]
d = (st[:, 1:] - st[:, :-1])
for i, n in enumerate(names):
    if i < d.shape[1]:
        col = d[:, i]
        print(f"{n:45s} median {int(col.median()):7d}  min {int(col.min()):7d}  max {int(col.max()):7d} cycles")
print("total entry -> last stamp: median", int((st[:, len(names)] - st[:, 0]).median()), "cycles")
print("per phase, summed over the four games (median workgroup):",
      {k: int(sum(d[:, 1 + 4 * g + j].median() for g in range(4))) for j, k in enumerate(["loop top", "draw", "wait B1", "tiles"])})
