#!/usr/bin/env python3
"""tg_step_tracked_i8 (the step that reads only what the action touches; nnz carried) against tg_step_i8, in place, on a
demo schedule that returns every game to its start (bench.py's): correctness of state / done / nnz against the full step,
then hipGraph-replayed time per launch.     python tools/tracked_time.py [S B K ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
args = [int(x) for x in sys.argv[1:]] or [25, 4096, 64, 25, 32768, 32, 16, 8192, 64, 16, 131072, 32]
for S, B, K in zip(args[0::3], args[1::3], args[2::3]):
    R = 8
    start, sched, _ = bench.make_demo_schedule(B, S, R, dev, 1, 0)
    L = len(sched)
    K = max(L, (K // L) * L)
    # correctness: one full cycle, tracked against full, step by step
    a, b = ops.alloc_states(B, S, dev), ops.alloc_states(B, S, dev)
    a.copy_(start)
    b.copy_(start)
    _, nnz = ops.done(a, want_nnz=True)
    ok = True
    for k in range(L):
        _, d_full = ops.step(a, sched[k], out=a)
        _, d_tr = ops.step_tracked(b, sched[k], nnz)
        _, n_full = ops.done(a, want_nnz=True)
        ok = ok and bool(torch.equal(a, b)) and bool(torch.equal(d_full, d_tr)) and bool(torch.equal(n_full, nnz))
    done = torch.zeros(B, dtype=torch.uint8, device=dev)
    pos = [0]

    def full():
        ops.step(a, sched[pos[0] % L], out=a, done=done)
        pos[0] += 1

    def tracked():
        ops.step_tracked(b, sched[pos[0] % L], nnz, done=done)
        pos[0] += 1

    pos[0] = 0
    t_full = bench.graph_time(full, dev, reps=K)
    pos[0] = 0
    t_tr = bench.graph_time(tracked, dev, reps=K)
    print(f"S={S} B={B}: ok={ok}  full {t_full * 1e6:8.2f} us  tracked {t_tr * 1e6:8.2f} us  ({t_full / t_tr:.2f}x)  "
          f"back at start: {bool(torch.equal(b, start))}", flush=True)
    del a, b
    torch.cuda.empty_cache()
