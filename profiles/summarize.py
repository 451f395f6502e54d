#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by profiles/run_profiles.sh) into the small files
committed under profiles/:
  <tag>_<run>_kernel_stats.csv    rocprofv3's own --stats table (averages; inflated for us-scale dispatches)
  <tag>_<run>_kernel_summary.json per kernel: count, min / median / mean duration and the median START-TO-START
                                  period of back-to-back dispatches (= duration + boundary, what HIP events see)
  <tag>_<run>.json                the bench JSON line of that (profiled or unprofiled) run
  traffic_<tag>.json              per-launch HBM bytes of the step kernels from the PMC passes
                                  (FETCH_SIZE doubled on gfx950, WRITE_SIZE as is; MI355X_MICROARCH.md, HBM)
  <tag>_mfma_pmc.json             SQ counters of the matrix-core kernels
  <tag>_launch_floor.txt          tools/microbench_step at BASELINE config 2 (empty / copy / step variants)
"""
import collections
import csv
import glob
import json
import shutil
import statistics
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = Path(__file__).resolve().parent.parent
src = root / "gpurun_out" / f"prof_{tag}"
dst = root / "profiles"


def find(dirpath, suffix):
    hits = sorted(Path(dirpath).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def kernel_summary(trace_csv):
    rows = []
    for r in csv.DictReader(open(trace_csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    rows.sort()
    dur, period = collections.defaultdict(list), collections.defaultdict(list)
    for i, (s, e, n) in enumerate(rows):
        dur[n].append(e - s)
        if i + 1 < len(rows) and rows[i + 1][2] == n and rows[i + 1][0] - s < 10 * (e - s):  # back-to-back only
            period[n].append(rows[i + 1][0] - s)
    out = {}
    for n, d in dur.items():
        out[n] = {"count": len(d), "min_us": round(min(d) / 1e3, 3), "median_us": round(statistics.median(d) / 1e3, 3),
                  "mean_us": round(sum(d) / len(d) / 1e3, 3),
                  "median_start_to_start_us": round(statistics.median(period[n]) / 1e3, 3) if period[n] else None}
    return out


for name in ("bench_driver", "bench_graph", "bench_graph_S16", "bench_eager", "generator"):
    f = find(src / name, "_kernel_stats.csv")
    if f:
        shutil.copy(f, dst / f"{tag}_{name}_kernel_stats.csv")
    t = find(src / name, "_kernel_trace.csv")
    if t:
        json.dump(kernel_summary(t), open(dst / f"{tag}_{name}_kernel_summary.json", "w"), indent=1, sort_keys=True)
for name in ("bench_driver", "bench_graph", "bench_graph_S16", "bench_eager", "bench_driver_cmd", "bench_default_lean"):
    j = src / f"{name}.json"
    if j.exists() and j.stat().st_size:
        shutil.copy(j, dst / f"{tag}_{name}.json")
for extra, name in (("bench_also_driver_cmd.json", "bench_also_driver_cmd.json"), ("fused_n1_n2.jsonl", "fused_n1_n2.jsonl")):
    if (src / extra).exists():
        shutil.copy(src / extra, dst / f"{tag}_{name}")
lf = src / "launch_floor.txt"
if lf.exists():
    shutil.copy(lf, dst / f"{tag}_launch_floor.txt")
if (src / "share_floor.txt").exists():
    shutil.copy(src / "share_floor.txt", dst / f"{tag}_share_floor.txt")
for probe, name in (("read_bw_probe", "read_bandwidth"), ("issue_rate_probe", "issue_rates"), ("shader_clock_probe", "shader_clock"),
                    ("expand_probe", "expand_probe"), ("aux_ops", "aux_ops"), ("generator_series", "generator_series"),
                    ("step_series", "step_series"), ("stride_ab", "stride_ab"), ("expand25_probe", "expand25_probe"),
                    ("stream_share", "stream_share"), ("stream_small", "stream_small")):
    if (src / f"{probe}.txt").exists():
        shutil.copy(src / f"{probe}.txt", dst / f"{tag}_{name}.txt")

traffic = {}
for d in sorted(glob.glob(str(src / "pmc_*_SIZE"))):
    parts = Path(d).name.split("_")  # pmc S4 B65536 FETCH SIZE
    key, ctr = f"{parts[1]}_{parts[2]}", parts[3] + "_SIZE"
    f = find(d, "_counter_collection.csv")
    if not f:
        continue
    vals, durs = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        # the STEP kernels: s4_kernel<0>, packed_kernel<S, TS, 0>, s16_step_kernel<0, LINES>, s25_step_kernel, s9_step_kernel
        if "tg::" in n and ("<0>" in n or ", 0>" in n or "<0, " in n or "s25_step_kernel" in n or "s9_step_kernel" in n
                            or "s4_step_kernel" in n) and "copy" not in n and "step_emit" not in n:
            vals[n].append(float(r["Counter_Value"]))
            durs[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for n, v in vals.items():
        e = traffic.setdefault(key, {"kernel": n.split("(")[0].replace("void ", ""), "launches": len(v), "round": tag})
        kb = sum(v) / len(v)
        e[ctr + "_KB_per_launch"] = kb
        e[ctr + "_avg_kernel_us_under_pmc"] = round(sum(durs[n]) / len(durs[n]) / 1e3, 2)
for key, e in traffic.items():
    if "FETCH_SIZE_KB_per_launch" in e and "WRITE_SIZE_KB_per_launch" in e:
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"]) * 1024)
        S, B = int(key.split("_")[0][1:]), int(key.split("_")[1][1:])
        e["algorithmic_bytes_per_launch"] = B * (2 * S ** 3 + 3 * S + 1)
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 4)
if traffic:
    json.dump(traffic, open(dst / f"traffic_{tag}.json", "w"), indent=1, sort_keys=True)

# PMC counters of the matrix-core kernels (S=25, B=4096, R=K=64): averages per launch
mfma = {}
for d in sorted(glob.glob(str(src / "mfma_*_p?"))):
    f = find(d, "_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "mfma" in n or "gen_" in n or "genfused" in n:
            agg[(n.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kern, ctr), v in agg.items():
        mfma.setdefault(kern, {})[ctr] = round(sum(v) / len(v), 1)
for kern, e in mfma.items():
    if "SQ_WAVES" in e and "SQ_INSTS_VALU" in e:
        e["valu_per_wave"] = round(e["SQ_INSTS_VALU"] / e["SQ_WAVES"], 1)  # a wavefront serves several games
        e["valu_per_game_and_wavefront"] = round(e["SQ_INSTS_VALU"] / (4096 * 4), 1)  # B = 4096, 4 wavefronts per game
if mfma:
    json.dump(mfma, open(dst / f"{tag}_mfma_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1, sort_keys=True))
