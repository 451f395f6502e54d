"""bench.py pieces that do not need a GPU: the algorithmic byte counts of SURVEY.md section 8(d), the
roofline object's arithmetic (no fraction above 1 from skipped stores), the committed traffic file it reads,
and the N>1 self-launch path (two gloo ranks, --dry-run)."""
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # defines functions only; main() runs under __main__
    return mod


def test_algorithmic_bytes_per_step(bench):
    # 2 S^3 + 3 S + 1  ->  141 / 8 241 / 31 326 bytes
    assert [bench.bytes_step(S) for S in (4, 16, 25)] == [141, 8241, 31326]


def test_roofline_object(bench):
    # 9 samples of 1000 launches around 2.5 ms -> 2.5 us per launch (median)
    r = bench.roofline(65536, 4, 1000, [2.5] * 5 + [2.4, 2.6, 9.0, 2.5], None, (4200.0, 2.2))
    assert r["bound"] == "launch" and r["unit"] == "GB/s" and r["peak"] == 8000.0   # 4 MiB of states: L2-resident
    assert r["bytes_per_launch"] == r["needed_bytes_per_launch"] == 65536 * 141
    assert abs(r["avg_launch_us"] - 2.5) < 1e-9 and len(r["launch_us_samples"]) == 9
    assert abs(r["achieved"] - 65536 * 141 / 2.5e-6 / 1e9) < 0.01
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4 and r["frac"] == r["frac_algorithmic"]
    assert r["copy_ceiling_GBps"] == 4200.0 and abs(r["frac_of_copy_ceiling"] - r["achieved"] / 4200.0) < 1e-3
    assert r["regime"].startswith("L2-resident") and "hbm_copy_ceiling_GBps" not in r
    # traffic comes from the committed PMC summary of the newest round that recorded this kernel
    if r["traffic"] is not None:
        t = json.loads((ROOT / "profiles" / f"traffic_{r['traffic_round']}.json").read_text())
        assert r["traffic"] == t["S4_B65536"]["hbm_bytes_per_launch"]
        assert 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.05  # S=4 moves what the algorithm needs, no more
        assert abs(r["frac_traffic"] - r["traffic"] / 2.5e-6 / 1e9 / 8000.0) < 1e-3


def test_regimes_by_footprint(bench):
    """Only footprints of 2 GiB and more are labelled (and bounded) as HBM streams; everything the caches can hold or
    assist says so (256-512 MiB 'HBM' lines moved more bytes per second than HBM delivers)."""
    labels = [(bench.regime_of(n)[1], bench.regime_of(n)[0].split(":")[0]) for n in
              (4 << 20, 31 << 20, 32 << 20, 255 << 20, 256 << 20, 512 << 20, (2 << 30) - 1, 2 << 30, 4 << 30)]
    assert labels == [("launch", "L2-resident"), ("launch", "L2-resident"), ("cache", "Infinity-Cache-resident"),
                      ("cache", "Infinity-Cache-resident"), ("cache", "cache-assisted"), ("cache", "cache-assisted"),
                      ("cache", "cache-assisted"), ("hbm", "hbm-streaming"), ("hbm", "hbm-streaming")]
    # an HBM line carries the copy ceiling measured in the run and its fraction of it
    r = bench.roofline(1 << 25, 4, 16, [0.82 * 16] * 3, None, (6700.0, 640.0), hbm_copy=(6700.0, 640.0))
    assert r["bound"] == "hbm" and r["hbm_copy_ceiling_GBps"] == 6700.0 and r["hbm_achievable_GBps_guide"] == 6300.0
    assert abs(r["frac_of_hbm_copy_ceiling"] - r["achieved"] / 6700.0) < 1e-3 and r["frac"] < r["frac_of_hbm_copy_ceiling"]
    r = bench.roofline(8192, 16, 512, [6.0e-3 * 512] * 3, None, None, hbm_copy=(6700.0, 640.0))
    assert r["bound"] == "cache" and "frac_of_hbm_copy_ceiling" not in r


def test_no_fraction_above_one_from_skipped_stores(bench):
    """Round 1 reported 1.21 for S=16, B=8192: algorithmic bytes / time with the in-place kernel skipping 91 % of
    its stores.  `frac` now prices the NEEDED bytes (exact, from the schedule), which stay below the peak at the
    measured 7.0 us per launch; the algorithmic figure is kept beside it under its own name."""
    B, S = 8192, 16
    g = torch.Generator().manual_seed(0)
    tok = torch.multinomial(torch.tensor([0.15, 0.7, 0.15]), B * 3 * S, replacement=True, generator=g)
    tok = tok.to(torch.int8).reshape(B, 3 * S)
    need = bench.needed_bytes_per_launch(B, S, [tok])
    t = tok.to(torch.int64) - 1
    rows = ((t[:, :S] != 0).sum(1) * (t[:, S:2 * S] != 0).sum(1) * (t[:, 2 * S:] != 0).any(1)).sum().item()
    assert need == B * (S ** 3 + 3 * S + 1) + 16 * rows        # S=16: a chunk is a row (i, j)
    assert need < 0.6 * B * bench.bytes_step(S)                 # ~9 % of the rows change
    r = bench.roofline(B, S, 512, [7.0e-3 * 512] * 5, need)
    assert r["frac_algorithmic"] > 1.0 and r["frac"] < 1.0 and r["frac"] == round(need / 7.0e-6 / 8e12, 4)
    # S=25: chunks straddle rows; S=4 always stores
    tok25 = torch.ones((3, 75), dtype=torch.int8)
    tok25[0, [0, 25, 50]] = 2                                  # one element (0,0,0) -> one chunk
    tok25[1, :] = 2                                            # dense action: all 977 chunks
    assert bench.changed_chunks_per_launch(25, [tok25]) == 1 + 977
    assert bench.needed_bytes_per_launch(7, 4, [torch.ones((7, 12), dtype=torch.int8)]) == 7 * 141


def test_stale_traffic_entry_is_omitted(bench):
    assert bench.measured_traffic(65536, 4, "tg::some_other_kernel<0>") == (None, None)


def _last_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert lines and lines[-1].startswith("{"), stdout[-500:]     # the contract line is the LAST line of stdout
    return lines[-1]


def test_self_launch_two_ranks_dry_run(tmp_path, bench):
    """`python bench.py --gpus 2` with no launcher in the environment starts its two ranks itself and relays
    rank 0's line (dry run: rendezvous over gloo + shard arithmetic, no GPU).  Default at N>1: weak scaling of the
    N=1 workload (65 536 games per GPU); the line also names the shards of the extras an N>1 run measures."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert res.returncode == 0, res.stderr[-3000:]
    line = _last_line(res.stdout)
    assert len(line) < bench.MAX_LINE_BYTES
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 * 65536 and out["config"]["batch_rank0"] == 65536
    assert out["config"]["last_game_id"] == 2 * 65536 and "config 2" in out["config"]["workload"]
    sh = out["shards"]
    assert sh["s16_weak"] == {"S": 16, "global_batch": 16384, "batch_rank0": 8192}
    assert sh["s16_strong"] == {"S": 16, "global_batch": 8192, "batch_rank0": 4096}
    assert sh["s4_strong"] == {"S": 4, "global_batch": 1 << 20, "batch_rank0": 1 << 19}
    # strong scaling on request: BASELINE config 4, 2^20 games in total
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--scaling", "strong", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(_last_line(res.stdout))
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 1 << 20 and out["config"]["batch_rank0"] == 1 << 19
    assert "config 4" in out["config"]["workload"]


def _full_object(bench, world):
    """A headline object as main() assembles it, from canned numbers, with every optional field present and the
    longest kernel names / workload strings the bench can produce."""
    res = {"lead_in": 8, "run_ms_with_lead_in": [0.08] * 9, "run_ms_lead_in_only": [0.03] * 9}
    B = 65536
    r4 = bench.roofline(B, 4, 20, [2.3e-3 * 20] * 9, None, (4200.123, 2.234), res=res, hbm_copy=(6712.3, 640.12))
    r16 = bench.roofline(1 << 17, 16, 512, [98.24e-3 * 512] * 5, 612345678.9, (5321.1, 201.2), hbm_copy=(6712.3, 640.12))
    assert r16["kernel"] == "tg::s16_step_kernel<0, true, true, true>"
    full = {"metric": "env steps/sec (batched games)", "value": 19612345678.9, "unit": "steps/s", "n_gpus": world, "steps": 2016,
            "warmup": 224, "samples": 9, "ms_per_step": 0.003338, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": "BASELINE config 4: S=4 int8, 1048576 games in total sharded over 8 GPUs (131072 per GPU); one "
                                   "in-place tg_step_i8 launch per step", "S": 4, "batch_per_gpu": B, "global_batch": B * world,
                       "launch": "graph", "parallelism": f"shard{world} (contiguous game ranges, no collective)",
                       "timing": "x" * 300, "wall_us_per_sample": [66.123] * 9},
            "roofline": r4,
            "cpu_baseline": {"value": 5412345.6, "unit": "steps/s", "cores": 16, "kind": "port", "c_int8_port_1thread_steps_per_s": 1.0,
                             "sample": "63 batched steps of the same workload (B=65536, S=4) in 12.0 s", "sample_detail": "y" * 400,
                             "host_cpus_available": 256, "per_game_loop_steps_per_s": 19812.3},
            "value_s16": 1540819744.3, "ms_per_step_s16": 0.00526, "roofline_s16": r16,
            "also_file": "bench_also.json", "also_ok": {"entries": 48, "failed": []}}
    fig = {"global_batch": 1 << 20, "batch_per_gpu": 131072, "value": 241234567890.1, "ms_per_step": 0.004351,
           "launch_us": 3.335, "event_steps_per_s": 314123456789.0}
    if world > 1:
        full["per_rank"] = {"event_launch_us_max_over_ranks": 2.4}
        full["s16_strong"] = dict(fig)
        full["s4_weak"] = dict(fig)
        full["s4_strong"] = dict(fig, one_gpu_launch_us=21.523, speedup_event=6.453, speedup_wall=6.012, ideal=8)
        full["streamed_s4"] = {"global_batch": 1 << 20, "share_us_per_step": 0.805, "share_mode": "ready", "one_gpu_us_per_step": 5.221,
                               "one_gpu_mode": "rounds", "ok": True, "speedup": 6.486, "value": 1302579710144.9}
    else:
        full["cfg4_one_gpu"] = {"share_of_8_launch_us": 3.335, "whole_launch_us": 21.523, "streamed_share_of_8_us_per_step": 0.805,
                                "streamed_whole_us_per_step": 5.221, "predicted_speedup_8gpu_event": 6.453,
                                "predicted_speedup_8gpu_streamed": 6.486}
        full["generator_cfg5"] = {"us_per_launch": 26.03, "demos_per_s": 157536950.5, "bound": "valu", "valu_issue_frac": 0.5774}
    return full


@pytest.mark.parametrize("world", [1, 8])
def test_contract_line_is_short_and_complete(bench, world):
    """round 3's line was 28.7 KB (40 `also` entries inline), the driver kept a tail of it and parsed nothing.
    The line is now built by contract_line(): contract fields + numeric roofline / cpu_baseline / S=16 / sharded objects,
    bounded at 4 KB; prose, sample lists and `also` go to bench_also.json."""
    full = _full_object(bench, world)
    line = bench.contract_line(full)
    assert len(line) < bench.MAX_LINE_BYTES == 4096 and "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "value_s16", "ms_per_step_s16", "roofline_s16"):
        assert k in out, k
    assert set(out["roofline"]) <= set(bench.ROOFLINE_KEYS) and set(out["roofline_s16"]) <= set(bench.ROOFLINE_KEYS)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"] and k in out["roofline_s16"]
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-4
    assert set(out["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample", "host_cpus_available", "per_game_loop_steps_per_s"}
    assert "timing" not in out["config"] and "wall_us_per_sample" not in out["config"] and "workload" in out["config"]
    assert "per_rank" not in out and "also" not in out
    if world > 1:
        assert out["s4_strong"]["speedup_event"] == 6.453 and out["streamed_s4"]["speedup"] == 6.486
    # a line that would not fit is refused, never printed
    full["config"]["workload"] = "w" * 5000
    with pytest.raises(RuntimeError):
        bench.contract_line(full)


def test_side_file_holds_what_left_the_line(bench, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    full = _full_object(bench, 1)
    name = bench.write_side_file(full, [{"workload": "x", "ok": True}])
    doc = json.loads((tmp_path / name).read_text())
    assert doc["headline"]["roofline"]["method"].startswith("HIP events") and doc["also"][0]["workload"] == "x"
    assert "launch_us_samples" in doc["headline"]["roofline"] and "timing" in doc["headline"]["config"]


def test_wrong_world_size_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr


def test_step_kernel_names_exist_in_the_library():
    """bench.step_kernel_name says which kernel tg_step_i8 launches for a workload (the roofline object quotes it and
    profiles/traffic_rNN.json is matched by it).  Every name it can return must be a kernel of the built library -- a
    renamed template parameter (round 3 added the digit-form flag) would otherwise silently drop `frac_traffic`."""
    import subprocess

    import bench
    from mat_mul_amd import build

    lib = build.build()
    nm = subprocess.run(["nm", "-C", str(lib)], capture_output=True, text=True)
    if nm.returncode != 0:
        pytest.skip("nm not available")
    MiB = 1 << 20
    names = set()
    for S, game in ((4, 64), (16, 4096), (25, 15632)):
        for nbytes in (1 * MiB, 100 * MiB, 330 * MiB, 400 * MiB, 1300 * MiB, 2100 * MiB):
            names.add(bench.step_kernel_name(S, max(1, nbytes // game)))
    assert len(names) >= 9
    for name in names:
        assert f"void {name}(" in nm.stdout, name
