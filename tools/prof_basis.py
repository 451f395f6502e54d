#!/usr/bin/env python3
"""The generator of BASELINE config 5 (per-GPU share: 4 096 demos, S=25, R=64) a few times, plain and in a random
basis -- for `rocprofv3 --kernel-trace --stats` (profiles/run_profiles.sh)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402

S, B, R = 25, 4096, 64
P = ops.sample_basis(B, S, "cuda:0", seed=3)
tok = torch.empty((B, R, 3 * S), dtype=torch.int8, device="cuda:0")
tgt = ops.alloc_states(B, S, "cuda:0")
for _ in range(5):
    ops.gen_demos(B, S, R, "cuda:0", seed=1, target=tgt, actions=tok)
torch.cuda.synchronize()
for _ in range(5):
    ops.gen_demos(B, S, R, "cuda:0", seed=1, basis=P, target=tgt, actions=tok)
torch.cuda.synchronize()
print("ok")
