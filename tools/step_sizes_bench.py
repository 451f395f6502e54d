#!/usr/bin/env python3
"""The single-step kernel over batch sizes (bench.py's timing machinery):
    python tools/step_sizes_bench.py [--pad N] [--copy] S B K [S B K ...]
--pad N: game stride rounded up to N bytes (default 16); --copy: also time tg_copy_i8 on the same footprint.
A/B switches come from the environment (TG_LIB_VARIANT=ab TG_S16_LINES=1 ...), one process per variant."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

argv = sys.argv[1:]
pad, want_copy = 16, False
while argv and argv[0].startswith("--"):
    if argv[0] == "--pad":
        pad = int(argv[1])
        argv = argv[2:]
    elif argv[0] == "--copy":
        want_copy = True
        argv = argv[1:]
    else:
        raise SystemExit(f"unknown flag {argv[0]}")
dev = torch.device("cuda", 0)
args = [int(x) for x in argv] or [16, 8192, 512, 16, 32768, 256, 16, 131072, 64, 25, 4096, 208, 9, 32768, 256]
sw = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TG_"))
for s2, b2, k2 in zip(args[0::3], args[1::3], args[2::3]):
    st, sc, _ = bench.make_demo_schedule(b2, s2, 7 if s2 == 4 else 8, dev, 1, 0, pad_to=pad)
    tm = bench.StepTimer(st, sc, dev, "graph", pad_to=pad)
    r2 = tm.measure(k2, 8 if b2 * s2 ** 3 > (1 << 30) else 32, 5)
    ro = bench.roofline(b2, s2, k2, r2["event_ms"], bench.needed_bytes_per_launch(b2, s2, sc))
    stride = st.stride(0)
    line = (f"S={s2} B={b2} stride={stride} ({b2 * stride / 2 ** 20:.0f} MiB) [{sw}]: ok={r2['ok']} {ro['avg_launch_us']} us/launch  "
            f"frac={ro['frac']}  frac_algorithmic={ro['frac_algorithmic']}  needed {ro['needed_bytes_per_launch'] / 1e6:.1f} MB "
            f"-> {ro['achieved']} GB/s")
    del tm
    if want_copy:
        gbps, us = bench.copy_ceiling_gbps(b2, s2, dev, K=8 if b2 * s2 ** 3 > (1 << 27) else 64, pad_to=pad)
        line += f"  | copy {us} us = {gbps} GB/s"
    print(line, flush=True)
    del st, sc
    torch.cuda.empty_cache()
