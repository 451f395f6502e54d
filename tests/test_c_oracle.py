"""The plain-C oracle (oracle/tg_oracle.c) against the reference's golden fixtures and against the
numpy oracle: two independent restatements must agree everywhere.  CPU only."""
import numpy as np
import pytest

from oracle import tensor_game as O
from oracle.c_oracle import COracle


@pytest.fixture(scope="module")
def c():
    return COracle()


def test_c_oracle_golden(c, golden):
    g = golden("strassen")
    state = g["tensor"][None].astype(np.int8)
    for k in range(7):
        state, done, ovf = c.step_i8(state, g["tokens"][k][None])
        assert np.array_equal(state[0], g["replay"][k + 1]) and int(done[0]) == int(k == 6) and not ovf.any()
    new, done, _ = c.step_i8(g["ds_states"], g["ds_actions"], shift=2)
    assert np.array_equal(done.astype(bool), g["ds_rewards"] == -1)
    m = golden("matmul_tensors")
    for n in (2, 3, 4, 5):
        assert np.array_equal(c.matmul_tensor(n), m[f"n{n}_t1"][0])
    s = golden("step_cases")
    for tag in sorted({k.rsplit("_", 1)[0] for k in s.files if k.endswith("_state")}):
        kids, done, chg, ovf = c.expand_i8(s[tag + "_state"][:, 0], s[tag + "_actions"])
        assert np.array_equal(kids, s[tag + "_children"][:, :, 0]) and np.array_equal(done, s[tag + "_done"])
        assert np.array_equal(chg, s[tag + "_changed"]) and not ovf.any()
    d = golden("synthetic_demos")
    for nm in sorted(k[: -len("_tokens")] for k in d.files if k.endswith("_tokens") and "_item" not in k):
        tgt, ovf = c.gen_from_factors_i8(d[nm + "_tokens"][None])
        assert np.array_equal(tgt[0], d[nm + "_target"]) and not ovf.any()


def test_c_oracle_equals_numpy_oracle(c):
    rng = np.random.default_rng(7)
    assert c.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    for S, B, K in [(1, 3, 2), (4, 50, 7), (5, 9, 4), (9, 12, 6), (16, 5, 9), (25, 2, 5), (32, 1, 3)]:
        st = rng.integers(-128, 128, size=(B, S, S, S)).astype(np.int8)
        st[::2] = rng.integers(-2, 3, size=st[::2].shape)
        ac = rng.integers(-3, 6, size=(B, K, 3 * S)).astype(np.int8)
        ac[::3] = rng.integers(0, 3, size=ac[::3].shape)
        for shift in (1, 2):
            for got, want in zip(c.step_i8(st, ac[:, 0], shift), O.step_i8(st, ac[:, 0], shift)):
                assert np.array_equal(got, want)
            for got, want in zip(c.step_many_i8(st, ac, shift), O.step_many_i8(st, ac, shift)):
                assert np.array_equal(got, want)
            for got, want in zip(c.expand_i8(st, ac, shift), O.expand_i8(st, ac, shift)):
                assert np.array_equal(got, want)
            for got, want in zip(c.gen_from_factors_i8(ac, shift), O.gen_from_factors_i8(ac, shift)):
                assert np.array_equal(got, want)
        assert np.array_equal(c.state_hash(st), O.state_hash(st))
    for S, R, probs, values in [(4, 7, (0.15, 0.7, 0.15), (-1, 0, 1)), (9, 5, (0.1, 0.8, 0.1), (-1, 0, 1)),
                                (25, 3, (1, 2, 10, 2, 1), (-2, -1, 0, 1, 2))]:
        thr = O.categorical_thresholds(probs)
        for got, want in zip(c.gen_demos_i8(11, S, R, thr, values, 1, seed=99, game_id_offset=2 ** 33 + 5),
                             O.gen_demos_i8(11, S, R, thr, values, 1, seed=99, game_id_offset=2 ** 33 + 5)):
            assert np.array_equal(got, want)
