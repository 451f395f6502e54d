#!/usr/bin/env python3
"""Kernel time of the byte-streaming entries either side of the step (expand, done, hash, reset, model-input frames):
each call captured `reps` times in a hipGraph and replayed (bench.graph_time), so the host's ~10 us per ctypes call is
not in the number (tools/bench_ops.py times eager calls: anything below ~13 us there is the host).  Prices the bytes a
call moves against 8 TB/s.   python tools/aux_time.py [--only emit]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--T", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda", 0)


def line(name, S, B, sec, nbytes):
    print(f"{name:28s} S={S:2d} B={B:8d}: {sec * 1e6:8.2f} us  {nbytes / sec / 1e9:7.0f} GB/s  frac {nbytes / sec / 8e12:.2f}", flush=True)


for S, B, R in [(4, 65536, 7), (4, 1 << 20, 7), (9, 32768, 12), (16, 8192, 20), (25, 4096, 64)]:
    N = S ** 3
    tokens, target = ops.gen_demos(B, S, R, dev, seed=1)
    reps = 20 if B * N < (32 << 20) else 5
    if not args.only or args.only == "emit":
        T = args.T
        ring = ops.alloc_ring(B, S, T, dev)
        for f in range(T):
            ring[:, f].copy_(target)
        sc = torch.empty((B, 1), dtype=torch.float32, device=dev)
        for dt, w in [(torch.float32, 4), (torch.float16, 2), (torch.bfloat16, 2)]:
            x = torch.empty((B, T, S, S, S), dtype=dt, device=dev)
            sec = bench.graph_time(lambda: ops.emit_frames(ring, 1, 1.0, dt, out=x, scalars=sc), dev, reps=reps)
            line(f"emit_frames {str(dt)[6:]} T={T}", S, B, sec, B * (T * N * (1 + w) + 4))
            del x
        del ring
    if not args.only or args.only == "expand":
        k = 8
        kids = ops.alloc_states(B * k, S, dev).unflatten(0, (B, k))
        kd = torch.zeros((B, k), dtype=torch.uint8, device=dev)
        kc = torch.zeros((B, k), dtype=torch.uint8, device=dev)
        ak = tokens[:, :k].contiguous() if R >= k else tokens[:, :1].expand(B, k, 3 * S).contiguous()
        sec = bench.graph_time(lambda: ops.expand(target, ak, out=kids, done=kd, changed=kc), dev, reps=reps)
        line("expand k=8", S, B, sec, B * (N + k * (N + 3 * S + 2)))
        del kids
    if not args.only or args.only == "small":
        state = ops.alloc_states(B, S, dev)
        state.copy_(target)
        sec = bench.graph_time(lambda: ops.done(state, want_nnz=True), dev, reps=reps)
        line("done + nnz", S, B, sec, B * (N + 5))
        sec = bench.graph_time(lambda: ops.state_hash(state), dev, reps=reps)
        line("state_hash", S, B, sec, B * (N + 8))
        start = target[0].contiguous()
        sec = bench.graph_time(lambda: ops.reset_broadcast(state, start), dev, reps=reps)
        line("reset_broadcast", S, B, sec, B * N)
