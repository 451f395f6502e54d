#!/usr/bin/env python3
"""Time every C-ABI entry point at the BASELINE config sizes (one MI355X), with HIP events on the
launch stream, and price each against its roofline:

  step / expand / done / reset : HBM bytes            step_many : HBM bytes (K fused steps)
  gen_from_factors / gen_demos : max(HBM write, integer MACs)   change_basis: integer MACs

Prints one JSON object per line.  Usage: python tools/bench_ops.py [--quick]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402

DEV = "cuda:0"
HBM = 8000.0  # GB/s
# integer VALU peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-ops/s (one v_mad_i32_i24 = 1 MAC)
VALU_GMACS = 256 * 4 * 32 * 2.4


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def emit(name, S, B, secs, nbytes=None, macs=None, units=None, **extra):
    row = {"op": name, "S": S, "B": B, "us": round(secs * 1e6, 2)}
    if units:
        row["units_per_s"] = round(units / secs, 1)
    if nbytes:
        row["GBps"] = round(nbytes / secs / 1e9, 1)
        row["hbm_frac"] = round(nbytes / secs / 1e9 / HBM, 4)
    if macs:
        row["GMACps"] = round(macs / secs / 1e9, 1)
        row["valu_frac"] = round(macs / secs / 1e9 / VALU_GMACS, 4)
    row.update(extra)
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    cfgs = [(4, 65536, 7), (4, 1 << 20, 7), (16, 8192, 20), (25, 4096, 64), (9, 32768, 12)]
    if args.quick:
        cfgs = [(4, 65536, 7), (16, 8192, 20), (25, 4096, 64)]
    for S, B, R in cfgs:
        N = S ** 3
        overflow = torch.zeros(B, dtype=torch.uint8, device=DEV)
        # generator
        tokens = torch.empty((B, R, 3 * S), dtype=torch.int8, device=DEV)
        target = ops.alloc_states(B, S, DEV)
        t = timeit(lambda: ops.gen_demos(B, S, R, DEV, seed=1, target=target, actions=tokens, overflow=overflow))
        emit("gen_demos (tokens + target)", S, B, t, nbytes=B * (N + 2 * 3 * S * R), macs=B * R * N, units=B, R=R,
             note="bytes = write S^3 + write and re-read 3SR tokens")
        t = timeit(lambda: ops.gen_from_factors(tokens, S, out=target, overflow=overflow))
        emit("gen_from_factors", S, B, t, nbytes=B * (N + 3 * S * R), macs=B * R * N, units=B, R=R)
        # single step, in place
        state = ops.alloc_states(B, S, DEV)
        state.copy_(target)
        done = torch.zeros(B, dtype=torch.uint8, device=DEV)
        a0 = tokens[:, 0].contiguous()
        t = timeit(lambda: ops.step(state, a0, out=state, done=done, overflow=overflow), iters=50)
        emit("step (in place)", S, B, t, nbytes=B * (2 * N + 3 * S + 1), units=B)
        # K fused steps
        state.copy_(target)
        ds = torch.zeros(B, dtype=torch.int32, device=DEV)
        t = timeit(lambda: ops.step_many(target, tokens, out=state, done_step=ds, overflow=overflow))
        emit("step_many", S, B, t, nbytes=B * (2 * N + R * 3 * S + 4), macs=B * R * N, units=B * R, K=R)
        # expand k=8
        k = 8
        kids = ops.alloc_states(B * k, S, DEV).unflatten(0, (B, k))
        kd = torch.zeros((B, k), dtype=torch.uint8, device=DEV)
        kc = torch.zeros((B, k), dtype=torch.uint8, device=DEV)
        ak = tokens[:, :k].contiguous() if R >= k else tokens[:, :1].expand(B, k, 3 * S).contiguous()
        t = timeit(lambda: ops.expand(target, ak, out=kids, done=kd, changed=kc))
        emit("expand k=8", S, B, t, nbytes=B * (N + k * (N + 3 * S + 2)), units=B * k)
        del kids
        # terminal check
        t = timeit(lambda: ops.done(state, want_nnz=True), iters=50)
        emit("done + nnz", S, B, t, nbytes=B * (N + 5), units=B)
        # reset
        import math
        n = math.isqrt(S)
        if n * n == S:
            t = timeit(lambda: ops.reset_matmul(state, n), iters=50)
            emit("reset_matmul", S, B, t, nbytes=B * N, units=B)
        start = target[0].contiguous()
        t = timeit(lambda: ops.reset_broadcast(state, start), iters=50)
        emit("reset_broadcast", S, B, t, nbytes=B * N, units=B)
        # next rows: model input, hash, rank
        T = 4
        ring = ops.alloc_ring(B, S, T, DEV)
        ring[:, 0].copy_(target)
        x32 = torch.empty((B, T, S, S, S), dtype=torch.float32, device=DEV)
        sc = torch.empty((B, 1), dtype=torch.float32, device=DEV)
        t = timeit(lambda: ops.emit_frames(ring, 0, 1.0, torch.float32, out=x32, scalars=sc))
        emit("emit_frames f32 T=4", S, B, t, nbytes=B * T * N * 5, units=B)
        x16 = torch.empty((B, T, S, S, S), dtype=torch.float16, device=DEV)
        t = timeit(lambda: ops.emit_frames(ring, 0, 1.0, torch.float16, out=x16, scalars=sc))
        emit("emit_frames f16 T=4", S, B, t, nbytes=B * T * N * 3, units=B)
        del x32, x16, ring
        t = timeit(lambda: ops.state_hash(target), iters=50)
        emit("state_hash", S, B, t, nbytes=B * (N + 8), units=B)
        Br = min(B, 16384)
        t = timeit(lambda: ops.slice_rank(target[:Br]))
        emit("slice_rank", S, Br, t, nbytes=Br * (N + 4), units=Br, macs=Br * 2 * S * S * S * S)
        # basis
        Bb = min(B, 8192)
        P = ops.sample_basis(Bb, S, DEV, seed=3)
        t = timeit(lambda: ops.sample_basis(Bb, S, DEV, seed=3))
        emit("sample_basis", S, Bb, t, units=Bb)
        P32 = P.to(torch.int32)
        src = target[:Bb]
        dst = ops.alloc_states(Bb, S, DEV)
        t = timeit(lambda: ops.change_basis(src, P32, out=dst))
        emit("change_basis", S, Bb, t, nbytes=Bb * (2 * N + 12 * S * S), macs=Bb * 3 * S ** 4, units=Bb)
        tb = torch.empty((Bb, R, 3 * S), dtype=torch.int8, device=DEV)
        t = timeit(lambda: ops.gen_demos(Bb, S, R, DEV, seed=1, basis=P, target=dst, actions=tb))
        emit("gen_demos with basis", S, Bb, t, macs=Bb * R * (N + 3 * S * S), units=Bb, R=R)


if __name__ == "__main__":
    main()
