#!/usr/bin/env python3
"""tg_step_emit against step + emit_frames, tg_expand_keyed_i8 against expand + state_hash (bench.py's fused_lines):
    python tools/fused_time.py [B ...]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
batches = tuple(int(x) for x in sys.argv[1:]) or (65536, 1 << 20)
for line in bench.fused_lines(dev, batches):
    print(json.dumps({k: v for k, v in line.items() if k != "workload"}), line["workload"][:60], flush=True)
