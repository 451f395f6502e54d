"""CPU-only: the oracle's Philox stream against published known-answer vectors, and the oracle
generator against the reference sampler's statistics (tests/golden/sampler_stats.npz)."""
import numpy as np
from scipy import stats

from oracle import tensor_game as O


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, out in kat:
        got = O.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert got.tolist() == out


def test_generator_matches_reference_distribution(golden):
    g = golden("sampler_stats")
    for S, probs, key in [(4, (0.15, 0.7, 0.15), "S4_p70"), (9, (0.15, 0.7, 0.15), "S9_p70"), (4, (0.1, 0.8, 0.1), "S4_p80")]:
        thr = O.categorical_thresholds(probs)
        f = O.gen_factors(2000, S, 2, thr, (-1, 0, 1), seed=99)
        assert (f != 0).any(axis=-1).all()                      # no zero vector survives (utils.py:229)
        ours = np.array([(f == v).sum() for v in (-1, 0, 1)])
        ref = g[key + "_value_counts"]
        chi2, p, _, _ = stats.chi2_contingency(np.stack([ours, ref]))
        assert p > 1e-3, (key, ours, ref, p)
        # analytic: P(value | vector non-zero)
        pz = probs[1] ** S
        expect0 = (probs[1] - pz) / (1 - pz)
        assert abs(ours[1] / ours.sum() - expect0) < 0.01
        # reference's joint rejection rate: accepted / attempts = (1 - pz)^3
        n_terms, attempts = g[key + "_terms_attempts"]
        assert abs(n_terms / attempts - (1 - pz) ** 3) < 0.03


def test_generator_is_sharding_invariant_and_seeded():
    thr = O.categorical_thresholds((0.15, 0.7, 0.15))
    tok, tgt, _ = O.gen_demos_i8(12, 4, 7, thr, (-1, 0, 1), 1, seed=5)
    a, ta, _ = O.gen_demos_i8(5, 4, 7, thr, (-1, 0, 1), 1, seed=5, game_id_offset=0)
    b, tb, _ = O.gen_demos_i8(7, 4, 7, thr, (-1, 0, 1), 1, seed=5, game_id_offset=5)
    assert np.array_equal(np.concatenate([a, b]), tok) and np.array_equal(np.concatenate([ta, tb]), tgt)
    tok2, _, _ = O.gen_demos_i8(12, 4, 7, thr, (-1, 0, 1), 1, seed=6)
    assert not np.array_equal(tok, tok2)


def test_basis_invariants():
    thr = O.categorical_thresholds((0.2, 0.6, 0.2))
    for S in (4, 9):
        P, L, U = O.sample_basis(6, S, thr, (-1, 0, 1), seed=3)
        dets = np.round(np.linalg.det(P.astype(np.float64))).astype(int)
        assert set(np.abs(dets).ravel().tolist()) == {1}                              # (iii) unimodular
        Pinv = O.unimodular_inverse(L, U)
        eye = np.broadcast_to(np.eye(S, dtype=np.int64), P.shape)
        assert np.array_equal(Pinv @ P, eye) and np.array_equal(P @ Pinv, eye)
        fthr = O.categorical_thresholds((0.15, 0.7, 0.15))
        tok, tgt, _ = O.gen_demos_i8(6, S, 5, fthr, (-1, 0, 1), 1, seed=8)
        tok_b, tgt_b, ovf_b = O.gen_demos_i8(6, S, 5, fthr, (-1, 0, 1), 1, seed=8, basis=P)
        moved, ovf = O.change_basis_i8(tgt, P)
        ok = (ovf == 0) & (ovf_b == 0)
        assert ok.any()
        assert np.array_equal(moved[ok], tgt_b[ok])                                   # (i) mode product of sum == sum of transformed terms
        back, _ = O.change_basis_i8(moved, Pinv)
        assert np.array_equal(back[ok], tgt[ok])                                      # (ii) inverse restores T
        ident, _ = O.change_basis_i8(tgt, eye)
        assert np.array_equal(ident, tgt)                                             # (iv) identity is a no-op
    big = np.full((1, 4, 4, 4), 100, np.int8)
    M = np.broadcast_to(2 * np.eye(4, dtype=np.int64), (1, 3, 4, 4))
    out, ovf = O.change_basis_i8(big, M)
    assert ovf.tolist() == [1] and int(out[0, 0, 0, 0]) == 800 - 768             # (v) wrap + flag
