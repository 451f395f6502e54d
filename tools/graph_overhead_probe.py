#!/usr/bin/env python3
"""Where does a short hipGraph replay spend its time?  (bench.py --steps 20: 3.15 us per launch against 2.52 us in a
2016-node graph.)  (1) event time per replay vs nodes per graph, replays enqueued back to back; (2) the same K
launches timed by EVENT-RECORD NODES inside the graph (hipEventRecord on the capturing stream through ctypes on
PyTorch's own libamdhip64), which excludes everything the runtime does around a replay."""
import ctypes as C
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
hip = C.CDLL(str(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"))
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

B, S, R = 65536, 4, 7
actions, target = ops.gen_demos(B, S, R, dev, seed=0)
sched = [actions[:, k].contiguous() for k in range(R)]
for k in range(R):
    a = actions[:, k].clone()
    a[:, :S] = 2 - a[:, :S]
    sched.append(a.contiguous())
state = ops.alloc_states(B, S, dev)
state.copy_(target)
done = torch.zeros(B, dtype=torch.uint8, device=dev)
launch = ops.prepare_step(state, sched, done, None)
L = len(sched)


def capture(n, with_events=False):
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    g = torch.cuda.CUDAGraph()
    evs = None
    with torch.cuda.graph(g, stream=side):
        if with_events:
            e0, e1 = C.c_void_p(), C.c_void_p()
            assert hip.hipEventCreate(C.byref(e0)) == 0 and hip.hipEventCreate(C.byref(e1)) == 0
            launch(0)  # one lead-in node, untimed
            rc0 = hip.hipEventRecord(e0, C.c_void_p(side.cuda_stream))
            for j in range(n):
                launch((1 + j) % L)
            rc1 = hip.hipEventRecord(e1, C.c_void_p(side.cuda_stream))
            evs = (e0, e1, rc0, rc1)
        else:
            for j in range(n):
                launch(j % L)
    cur.wait_stream(side)
    return g, evs


print("nodes per graph | us per replay | us per node (back-to-back replays, events around each replay)")
for n in (1, 2, 5, 10, 20, 28, 56, 112, 448, 2016):
    g, _ = capture(n)
    g.replay()
    torch.cuda.synchronize()
    reps = 9
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    g.replay()
    for i in range(reps + 1):
        evs[i].record()
        if i < reps:
            g.replay()
    torch.cuda.synchronize()
    t = statistics.median(evs[i].elapsed_time(evs[i + 1]) for i in range(reps)) * 1e3
    print(f"{n:6d} | {t:9.2f} | {t / n:7.3f}")

print("in-graph event-record nodes around K chained launches:")
for n in (20, 112, 2016):
    try:
        g, (e0, e1, rc0, rc1) = capture(n, with_events=True)
    except Exception as e:  # ROCm 7.2: an event recorded during capture cannot be timed (hipEventElapsedTime: 400)
        print(f"  capture with event-record nodes failed at K={n}: {type(e).__name__}")
        break
    if rc0 or rc1:
        print(f"  hipEventRecord during capture failed: rc {rc0} {rc1}")
        break
    vals = []
    for _ in range(7):
        g.replay()
        torch.cuda.synchronize()
        ms = C.c_float()
        rc = hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        vals.append((rc, ms.value * 1e3 / n))
    print(f"  K={n}: rc/us per launch {[(r, round(v, 3)) for r, v in vals]}")
