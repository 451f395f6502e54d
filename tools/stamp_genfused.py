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

# lane 0 of the last wavefront (a storing one): per game: loop top (B2 passed), stores issued, draw done, B1 passed, tiles done
st2 = ovf.view(torch.int64)[:480].reshape(20, 24)[:, 12:].cpu()
names2 = []
for g in range(2):
    names2 += [f"last wave g{g}: loop top -> stores of prev issued", f"last wave g{g}: draw", f"last wave g{g}: wait B1",
               f"last wave g{g}: tiles", f"last wave g{g}: tiles done -> next loop top (B2)"]
d2 = st2[:, 1:] - st2[:, :-1]
for i, n in enumerate(names2):
    if i < d2.shape[1]:
        col = d2[:, i]
        print(f"{n:55s} median {int(col.median()):7d}  min {int(col.min()):7d}  max {int(col.max()):7d} cycles")

# per stamped workgroup (0, 100, ..., 1900): start relative to the first, duration, and the heavy wavefront's phases
e = st[:, 0] - st[:, 0].min()
last = st[:, 1 + 4 * 2 + 1]
print("workgroup: start (cycles after the first) / entry -> final loop top / draw g0, tiles g0, draw g1, tiles g1 / last wave: stores g1, wait B1 g1")
for w in range(20):
    print(f"  wg {100 * w:5d}: {int(e[w]):7d} / {int(last[w] - st[w, 0]):6d} / {int(d[w, 2])}, {int(d[w, 4])}, {int(d[w, 6])}, {int(d[w, 8])} / {int(d2[w, 5])}, {int(d2[w, 7])}")
