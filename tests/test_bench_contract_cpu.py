"""bench.py pieces that do not need a GPU: the algorithmic byte counts of SURVEY.md section 8(d), the
roofline object's arithmetic, and the committed traffic file it reads."""
import importlib.util
import json
from pathlib import Path

import pytest

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
    r = bench.roofline(65536, 4, 1000, 2.5)  # 1000 launches in 2.5 ms -> 2.5 us per launch
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["bytes_per_launch"] == 65536 * 141
    assert abs(r["avg_launch_us"] - 2.5) < 1e-9
    assert abs(r["achieved"] - 65536 * 141 / 2.5e-6 / 1e9) < 0.01
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    # traffic comes from the committed PMC summary of the newest round
    t = json.loads(sorted((ROOT / "profiles").glob("traffic_r*.json"))[-1].read_text())
    assert r["traffic"] == t["S4_B65536"]["hbm_bytes_per_launch"]
    assert 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.05  # S=4 moves what the algorithm needs, no more
