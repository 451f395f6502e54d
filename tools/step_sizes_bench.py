#!/usr/bin/env python3
"""The single-step kernel over batch sizes (bench.py's timing machinery): python tools/step_sizes_bench.py S B K [S B K ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
args = [int(x) for x in sys.argv[1:]] or [16, 8192, 512, 16, 32768, 256, 16, 131072, 64, 25, 4096, 208, 9, 32768, 256]
for s2, b2, k2 in zip(args[0::3], args[1::3], args[2::3]):
    st, sc, _ = bench.make_demo_schedule(b2, s2, 7 if s2 == 4 else 8, dev, 1, 0)
    tm = bench.StepTimer(st, sc, dev, "graph")
    r2 = tm.measure(k2, 32, 5)
    ro = bench.roofline(b2, s2, k2, r2["event_ms"], bench.needed_bytes_per_launch(b2, s2, sc))
    print(f"S={s2} B={b2}: ok={r2['ok']} {ro['avg_launch_us']} us/launch  frac={ro['frac']}  frac_algorithmic={ro['frac_algorithmic']}")
