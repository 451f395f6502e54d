import os, sys
os.environ["TG_LIB_VARIANT"] = "stamps"
sys.path.insert(0, "/root/repo")
import torch
from mat_mul_amd import ops
S, B, R = 25, 4096, 64
dev = "cuda:0"
ovf = torch.zeros(B, dtype=torch.uint8, device=dev)
tok, tgt = ops.gen_demos(B, S, R, dev, seed=2, overflow=ovf)
out = ops.alloc_states(B, S, dev); ds = torch.zeros(B, dtype=torch.int32, device=dev)
ovf.zero_()
for _ in range(3): ops.step_many(tgt, tok, out=out, done_step=ds, overflow=ovf)
torch.cuda.synchronize()
st = ovf.view(torch.int64)[:512].reshape(16, 32).cpu()
d = st[:, 11:20] - st[:, 10:19]
names = ["tile0", "tile1", "emit01+..", "tile2", "tile3", "emit23+..", "tile4", "(none)", "emit4"]
for i, n in enumerate(names):
    print(n, int(d[:, i].median()), int(d[:, i].min()), int(d[:, i].max()))
print("game0 phases:", [int((st[:, i + 1] - st[:, i]).median()) for i in range(1, 8)])

print("g1: bound->before wa/tile lambda:", int((st[:, 23] - st[:, 22]).median()), " ->first tile:", int((st[:, 10] - st[:, 23]).median()),
      " last emit->atomics done:", int((st[:, 24] - st[:, 19]).median()), " ->B2 passed:", int((st[:, 25] - st[:, 24]).median()))
