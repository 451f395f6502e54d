#!/usr/bin/env python3
"""Run ONE entry point a few times (for rocprofv3 --pmc / --kernel-trace passes).
python3 tools/prof_one.py --op genf|many|step|expand|gen --S 25 --B 4096 --R 64 --iters 5"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="genf")
ap.add_argument("--S", type=int, default=25)
ap.add_argument("--B", type=int, default=4096)
ap.add_argument("--R", type=int, default=64)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--basis", action="store_true", help="--op gen: in a random GL(S,Z) basis")
a = ap.parse_args()
dev = "cuda:0"
tokens, target = ops.gen_demos(a.B, a.S, a.R, dev, seed=1)
state = ops.alloc_states(a.B, a.S, dev)
state.copy_(target)
done = torch.zeros(a.B, dtype=torch.uint8, device=dev)
ds = torch.zeros(a.B, dtype=torch.int32, device=dev)
a0 = tokens[:, 0].contiguous()
k = min(8, a.R)
kids = ops.alloc_states(a.B * k, a.S, dev).unflatten(0, (a.B, k)) if a.op == "expand" else None
ak = tokens[:, :k].contiguous()
P = ops.sample_basis(a.B, a.S, dev, seed=11) if a.basis else None
torch.cuda.synchronize()
for _ in range(a.iters):
    if a.op == "genf":
        ops.gen_from_factors(tokens, a.S, out=state)
    elif a.op == "many":
        ops.step_many(target, tokens, out=state, done_step=ds)
    elif a.op == "step":
        ops.step(state, a0, out=state, done=done)
    elif a.op == "expand":
        ops.expand(target, ak, out=kids)
    elif a.op == "gen":
        ops.gen_demos(a.B, a.S, a.R, dev, seed=1, target=state, actions=tokens, basis=P)
torch.cuda.synchronize()
print("ok")
