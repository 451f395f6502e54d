#!/usr/bin/env python3
"""A/B library only: the fused generator (S=25, R=64, B=4096) with phases switched off (TG_GF_ABLATE bit mask) and
with a forced number of workgroups per CU (TG_GF_WGS) -- where does a launch spend its time?  Each case runs in a
child process (the switches are read once per process)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from mat_mul_amd import ops
S, B, R = 25, 4096, 64
basis = sys.argv[2] == "1"
dev = "cuda:0"
P = ops.sample_basis(B, S, dev, seed=3) if basis else None
tok = torch.empty((B, R, 3 * S), dtype=torch.int8, device=dev)
tgt = ops.alloc_states(B, S, dev)
fn = lambda: ops.gen_demos(B, S, R, dev, seed=1, basis=P, target=tgt, actions=tok)
for _ in range(3): fn()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for _ in range(20): fn()
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 20)
print(f"{sorted(ts)[2]:.2f}")
'''
script = Path("/tmp/ablate_child.py")
script.write_text(CHILD)
cases = [("all phases", 0), ("no Philox", 1), ("no draw LDS writes", 2), ("no Philox, no draw writes", 3), ("no tiles", 4),
         ("no target store", 8), ("no token store", 16), ("no stores", 24), ("draw only (no tiles, no stores)", 28),
         ("tiles only", 27), ("nothing but barriers", 31), ("barriers, no draw loop", 95),
         ("launch + dispatch only", 32)]
for basis in ("0",):
    print(f"basis={basis}")
    for name, mask in cases:
        env = dict(os.environ, TG_LIB_VARIANT="ab", TG_GF_ABLATE=str(mask))
        out = subprocess.run([sys.executable, str(script), str(ROOT), basis], env=env, capture_output=True, text=True)
        print(f"  ablate={mask:2d} {name:34s} {out.stdout.strip() or out.stderr[-300:]} us")
    for nw in (4,):
        for wgs in ():
            env = dict(os.environ, TG_LIB_VARIANT="ab", TG_GF_NW=str(nw), TG_GF_WGS=str(wgs))
            out = subprocess.run([sys.executable, str(script), str(ROOT), basis], env=env, capture_output=True, text=True)
            print(f"  wavefronts per workgroup = {nw}, workgroups per CU = {wgs or 'occupancy API'}: {out.stdout.strip() or out.stderr[-300:]} us")
    for wgs in ():
        env = dict(os.environ, TG_LIB_VARIANT="ab", TG_GF_WGS=str(wgs))
        out = subprocess.run([sys.executable, str(script), str(ROOT), basis], env=env, capture_output=True, text=True)
        print(f"  workgroups per CU = {wgs}: {out.stdout.strip() or out.stderr[-300:]} us")
