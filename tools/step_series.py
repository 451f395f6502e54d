#!/usr/bin/env python3
"""tg_step_i8 in place, one hipGraph of K chained launches replayed back to back after an idle gap: per-replay time per
launch, i.e. how the rate moves while the GPU's clocks settle under sustained load.
    python tools/step_series.py S B K [S B K ...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
args = [int(x) for x in sys.argv[1:]] or [4, 65536, 1008, 4, 131072, 1008, 16, 8192, 512]
for S, B, K in zip(args[0::3], args[1::3], args[2::3]):
    st, sc, _ = bench.make_demo_schedule(B, S, 7 if S == 4 else 8, dev, 1, 0)
    tm = bench.StepTimer(st, sc, dev, "graph")
    K = (K // tm.L) * tm.L
    g = tm._graph(0, K)
    g.replay()
    torch.cuda.synchronize(dev)
    time.sleep(0.5)
    n = 60
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n + 1):
        evs[i].record()
        if i < n:
            g.replay()
    torch.cuda.synchronize(dev)
    per = [evs[i].elapsed_time(evs[i + 1]) * 1e3 / K for i in range(n)]
    ok = bool(torch.equal(tm.state, tm.start))
    print(f"S={S} B={B}: {K} launches per replay, ok={ok}; us per launch by replay after 0.5 s idle: " + " ".join(f"{p:.3f}" for p in per), flush=True)
    del tm, g
