import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from mat_mul_amd import ops
S, B, R = 25, 4096, 64
P = ops.sample_basis(B, S, "cuda:0", seed=3)
tok = torch.empty((B, R, 3 * S), dtype=torch.int8, device="cuda:0")
tgt = ops.alloc_states(B, S, "cuda:0")
for _ in range(3):
    ops.gen_demos(B, S, R, "cuda:0", seed=1, basis=P, target=tgt, actions=tok)
torch.cuda.synchronize()
print("ok")
