import sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np, time
from mat_mul_amd import ops
import bench
dev = torch.device("cuda", 0)
for S, B in [(25, 4096), (16, 8192), (9, 32768), (4, 65536)]:
    tok, tgt = ops.gen_demos(B, S, 12 if S < 25 else 64, dev, seed=2)
    sec = bench.graph_time(lambda: ops.slice_rank(tgt), dev, reps=5)
    print(f"slice_rank S={S} B={B}: {sec*1e6:.1f} us  ({B/sec/1e6:.2f} M games/s)")
