import sys, statistics
sys.path.insert(0, '.')
import torch, bench
from mat_mul_amd import ops
dev = torch.device('cuda', 0)
for S, B, R in [(25, 4096, 64), (16, 8192, 20), (9, 8192, 12)]:
    tok, tgt = ops.gen_demos(B, S, R, dev, seed=1)
    P = ops.sample_basis(B, S, dev, seed=3).to(torch.int32)
    out = ops.alloc_states(B, S, dev)
    sec = bench.graph_time(lambda: ops.change_basis(tgt, P, out=out), dev, reps=10)
    print(f"S={S} B={B}: {sec*1e6:.1f} us")
