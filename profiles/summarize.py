#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by profiles/run_profiles.sh) into the
small files committed under profiles/: kernel stats CSVs, the bench JSON lines of the profiled
runs, and traffic_<tag>.json = per-launch HBM bytes of the step kernels from the PMC passes
(FETCH_SIZE doubled on gfx950, WRITE_SIZE as is; /opt/skills/guides/MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"
for name in ("bench_graph", "bench_graph_S16", "bench_eager", "generator"):
    f = src / name / f"{name}_kernel_stats.csv"
    if f.exists():
        shutil.copy(f, dst / f"{tag}_{name}_kernel_stats.csv")
    j = src / f"{name}.json"
    if j.exists():
        shutil.copy(j, dst / f"{tag}_{name}.json")

traffic = {}
for d in sorted(glob.glob(str(src / "pmc_*_SIZE"))):
    parts = Path(d).name.split("_")  # pmc S4 B65536 FETCH SIZE
    key, ctr = f"{parts[1]}_{parts[2]}", parts[3] + "_SIZE"
    vals, durs = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(Path(d) / "pmc_counter_collection.csv")):
        n = r["Kernel_Name"]
        if "tg::" in n and ("<0>" in n or ", 0>" in n):  # the STEP kernels
            vals[n].append(float(r["Counter_Value"]))
            durs[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for n, v in vals.items():
        e = traffic.setdefault(key, {"kernel": n.split("(")[0].replace("void ", ""), "launches": len(v)})
        kb = sum(v) / len(v)
        e[ctr + "_KB_per_launch"] = kb
        e[ctr + "_avg_kernel_us_under_pmc"] = round(sum(durs[n]) / len(durs[n]) / 1e3, 2)
for key, e in traffic.items():
    if "FETCH_SIZE_KB_per_launch" in e and "WRITE_SIZE_KB_per_launch" in e:
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"]) * 1024)
        S, B = int(key.split("_")[0][1:]), int(key.split("_")[1][1:])
        e["algorithmic_bytes_per_launch"] = B * (2 * S ** 3 + 3 * S + 1)
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 4)
json.dump(traffic, open(dst / f"traffic_{tag}.json", "w"), indent=1, sort_keys=True)

# PMC counters of the matrix-core kernels (S=25, B=4096, R=K=64): averages per launch
mfma = {}
for d in sorted(glob.glob(str(src / "mfma_*_p?"))):
    op = Path(d).name.split("_")[1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(Path(d) / "pmc_counter_collection.csv")):
        n = r["Kernel_Name"]
        if "mfma_kernel" in n and ("genf" in n) == (op == "genf"):
            agg[(n.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, ctr), v in agg.items():
        mfma.setdefault(kern, {})[ctr] = round(sum(v) / len(v), 1)
for kern, e in mfma.items():
    if "SQ_WAVES" in e and "SQ_INSTS_VALU" in e:
        e["valu_per_wave"] = round(e["SQ_INSTS_VALU"] / e["SQ_WAVES"], 1)  # a wavefront serves several games
        e["valu_per_game_and_wavefront"] = round(e["SQ_INSTS_VALU"] / (4096 * 4), 1)  # B = 4096, 4 wavefronts per game
json.dump(mfma, open(dst / f"{tag}_mfma_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1, sort_keys=True))
