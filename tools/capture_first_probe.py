"""Probe: the very first calls of the matrix-core entries (which query occupancy once) made INSIDE a hipGraph capture."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from mat_mul_amd import ops
dev="cuda:0"
B,S,K=64,25,64
tok=torch.ones((B,K,3*S),dtype=torch.int8,device=dev)
st=ops.alloc_states(B,S,dev); out=ops.alloc_states(B,S,dev)
ds=torch.zeros(B,dtype=torch.int32,device=dev)
tgt=ops.alloc_states(B,S,dev)
side=torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    ops.step_many(st, tok, out=out, done_step=ds)          # first ever call: inside capture
    ops.gen_from_factors(tok, S, out=tgt)
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
print("capture-first ok", int(ds[0]), bool(out.any()), bool(tgt.any()))
