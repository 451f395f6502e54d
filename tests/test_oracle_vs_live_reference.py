"""When the reference is present (build container only), run ITS functions on fresh seeded inputs in a
subprocess and compare with the oracle -- a live check on top of the recorded fixtures.  CPU only; skipped
where /root/reference does not exist (the GPU box)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import tensor_game as O

REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="the reference only exists in the build container")

SCRIPT = r'''
import sys, numpy as np, torch
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import utils, act, datasets
out = {}
rng = np.random.default_rng(int(sys.argv[2]))
for case in range(12):
    S = int(rng.choice([2, 3, 4, 5, 9, 16]))
    B, T, k = int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 5))
    st = torch.from_numpy(rng.integers(-3, 4, size=(B, T, S, S, S)).astype(np.float32))
    ac = torch.from_numpy(rng.integers(0, 3, size=(B, k, 3 * S)).astype(np.int64))
    kids = act.get_child_states(st, ac)
    out[f"c{case}_state"], out[f"c{case}_actions"] = st.numpy(), ac.numpy()
    out[f"c{case}_kids"] = np.stack([c.numpy() for c in kids], axis=1)
    out[f"c{case}_done"] = np.array([[bool(utils.tensor_factorized(utils.get_head_state(c[b:b+1]))) for c in kids] for b in range(B)])
    out[f"c{case}_nonnull"] = np.array(utils.remove_null_actions(st, kids), np.int64)
    seq = [a for a in ac[0]]
    out[f"c{case}_taken"] = datasets.SyntheticDemoDataset._take_actions(seq, st[0, 0]).numpy()
    out[f"c{case}_a2t"] = utils.action_to_tensor(ac).numpy()
    out[f"c{case}_rank"] = np.array([utils.get_rank(st[b:b+1]) for b in range(B)])
for n in (2, 3, 4, 5):
    out[f"mm{n}"] = utils.build_matmul_tensor(2, n, n, n).numpy()
np.savez(sys.argv[1], **out)
'''


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_matches_live_reference(tmp_path, seed):
    script, npz = tmp_path / "ref.py", tmp_path / "ref.npz"
    script.write_text(SCRIPT)
    subprocess.run([sys.executable, str(script), str(npz), str(seed)], check=True, cwd=tmp_path, timeout=300)
    g = np.load(npz)
    for case in range(12):
        st, ac = g[f"c{case}_state"], g[f"c{case}_actions"]
        kids = np.stack(O.get_child_states(st, ac), axis=1)
        assert np.array_equal(kids, g[f"c{case}_kids"]), case
        done = np.stack([O.done_per_game(kids[:, i, 0]) for i in range(ac.shape[1])], axis=1)
        assert np.array_equal(done, g[f"c{case}_done"])
        assert O.remove_null_actions(st, [kids[:, i] for i in range(ac.shape[1])]) == g[f"c{case}_nonnull"].tolist()
        assert np.array_equal(O.take_actions(list(ac[0]), st[0, 0]), g[f"c{case}_taken"])
        assert np.array_equal(O.action_to_tensor(ac), g[f"c{case}_a2t"])
        assert np.array_equal(O.slice_rank_exact(st[:, 0]), g[f"c{case}_rank"])
        # the int8 build semantics agree with the reference wherever nothing overflows
        kids8, done8, chg8, ovf8 = O.expand_i8(st[:, 0].astype(np.int8), ac.astype(np.int8))
        assert not ovf8.any() and np.array_equal(kids8, g[f"c{case}_kids"][:, :, 0]) and np.array_equal(done8, g[f"c{case}_done"])
    for n in (2, 3, 4, 5):
        assert np.array_equal(O.build_matmul_tensor(2, n, n, n), g[f"mm{n}"])
