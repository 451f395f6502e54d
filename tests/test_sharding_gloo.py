"""N>1 control path on CPU: two `gloo` ranks, one process each (the GPU run uses the same
RankGroup over RCCL).  Checks the shard arithmetic, the barrier + max-over-ranks timing reduction
bench.py relies on, and that sharded generation (oracle stream, keyed by global game id)
reassembles to the single-rank result -- the property section 8(e) requires of the GPU generator."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys, json, time
sys.path.insert(0, os.environ["TG_ROOT"])
import numpy as np
from mat_mul_amd.sharding import RankGroup, shard_range
from oracle import tensor_game as O

g = RankGroup("gloo")
assert g.world == 2 and g.dist is not None
n_games, S, R = 37, 4, 5
lo, hi = shard_range(n_games, g.rank, g.world)
thr = O.categorical_thresholds((0.15, 0.7, 0.15))
tok, tgt, _ = O.gen_demos_i8(hi - lo, S, R, thr, (-1, 0, 1), 1, seed=77, game_id_offset=lo)
g.barrier()
time.sleep(0.05 * (g.rank + 1))                      # rank 1 is the slow one
wall, bad = g.max_over_ranks(0.05 * (g.rank + 1), 0.0)
assert abs(wall - 0.10) < 1e-9 and bad == 0.0       # every rank sees the slowest rank's time
w2, bad2 = g.max_over_ranks(1.0, 1.0 if g.rank == 1 else 0.0)
assert bad2 == 1.0                                   # a failure on any rank is seen by all
parts = [None, None]
g.dist.all_gather_object(parts, (lo, hi, tok.tobytes(), tgt.tobytes()))
if g.rank == 0:
    full_tok, full_tgt, _ = O.gen_demos_i8(n_games, S, R, thr, (-1, 0, 1), 1, seed=77)
    assert parts[0][1] == parts[1][0] and parts[0][0] == 0 and parts[1][1] == n_games
    assert b"".join(p[2] for p in parts) == full_tok.tobytes()
    assert b"".join(p[3] for p in parts) == full_tgt.tobytes()
    print("RANK0_OK")
g.close()
'''


def test_two_rank_gloo_control_path(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TG_ROOT=str(ROOT), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    assert "RANK0_OK" in outs[0][0]


def test_single_rank_group_is_identity():
    from mat_mul_amd.sharding import RankGroup
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        g = RankGroup("gloo")
        assert g.world == 1 and g.dist is None
        g.barrier()
        assert g.max_over_ranks(1.5, 0.0) == (1.5, 0.0)
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
