#!/usr/bin/env python3
"""Golden fixture for the tree filter of MCTS expansion, recorded by RUNNING THE REFERENCE's ``extend_tree``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tree.py      (build container only)

``act.extend_tree`` (/root/reference/act.py:115-216) is driven with a stand-in for the policy network -- an object
whose ``fwd_infer`` hands out pre-drawn candidate actions and records what it was asked -- so everything between the
network and the tree is the reference's own code: ``get_child_states`` (act.py:266-275), ``remove_null_actions``
(utils.py:191-194), the ``state_to_str`` keys (utils.py:164-169), the filter ``c not in new_mc_tree`` (act.py:188-195)
and the recording of the expanded state (act.py:209-211).  Stored per expansion ATTEMPT (one ``fwd_infer`` call), in
the order the reference made them:
    parent   int8 (S,S,S)   the state that was expanded (head frame)
    actions  int8 (k,3S)    the candidate tokens the stand-in returned
    n_kept   int            how many candidates survived both filters (0 = the reference asked again)
    kept     int8 (k,3S)    the surviving tokens, in order (``not_dupl_actions``), padded with -1
    commit   uint8          1 = this attempt's parent entered the tree afterwards (the last attempt of an expansion)
Nothing of the reference is copied; the ``.npz`` holds inputs and outputs only.
"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


class RecordingPolicy:
    """Stand-in for AlphaTensor: candidate actions come from a seeded pool; every request is recorded."""

    device = "cpu"

    def __init__(self, torch, S, k, rng, pool, p_fresh):
        self.torch, self.S, self.k, self.rng, self.pool, self.p_fresh = torch, S, k, rng, pool, p_fresh
        self.calls = []

    def fwd_infer(self, state, scalars):
        torch = self.torch
        assert state.shape[0] == 1 and scalars.shape == (1, 1)
        pick = self.rng.integers(0, len(self.pool), size=self.k)
        acts = np.stack([self.pool[i] for i in pick])                      # (k,3S), tokens with shift 1
        fresh = self.rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(self.k, 3 * self.S))
        use_fresh = self.rng.random(self.k) < self.p_fresh                  # some of the candidates are new draws
        acts = np.where(use_fresh[:, None], fresh, acts).astype(np.int64)
        self.calls.append((state[0, 0].numpy().astype(np.int8).copy(), acts.astype(np.int8).copy()))
        if len(self.calls) > 100000:
            raise RuntimeError("the expansion does not terminate")
        return torch.from_numpy(acts).view(1, self.k, 3 * self.S), None, torch.zeros(())


def record(torch, act, utils, S, T, k, n_sim, max_actions, seed, start, n_pool=5, p_fresh=0.25, inverses=False):
    rng = np.random.default_rng(seed)
    pool = []
    while len(pool) < n_pool:                                              # non-null actions the search keeps re-using
        a = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=3 * S).astype(np.int64)
        if all((a[x * S:(x + 1) * S] != 1).any() for x in range(3)):
            pool.append(a)
    if inverses:                                                           # the same terms with u negated: a step back to a
        for a in list(pool):                                               # state the search has already expanded
            b = a.copy()
            b[:S] = 2 - b[:S]
            pool.append(b)
    pool.append(np.ones(3 * S, np.int64))                                  # a null action (all factors zero)
    z = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=3 * S).astype(np.int64)
    z[S:2 * S] = 1                                                         # null through v = 0 only
    pool.append(z)
    model = RecordingPolicy(torch, S, k, rng, pool, p_fresh)
    root = torch.zeros((1, T, S, S, S))
    root[0, 0] = torch.from_numpy(start.astype(np.float32))
    tree, info = {}, {}
    rows = []
    for _ in range(n_sim):
        seen_calls = len(model.calls)
        keys_before = set(tree)
        tree, info = act.extend_tree(model, root, 0, max_actions, tree, info)
        new_keys = [s for s in tree if s not in keys_before]
        attempts = model.calls[seen_calls:]
        for a_i, (parent, acts) in enumerate(attempts):
            last = a_i == len(attempts) - 1
            kept = np.full((k, 3 * S), -1, np.int8)
            n_kept = 0
            if last:
                assert len(new_keys) == 1
                key = utils.state_to_str(torch.from_numpy(parent.astype(np.float32)))
                assert key == new_keys[0]
                surv = info[key][5][0].numpy().astype(np.int8)              # not_dupl_actions (1,k',3S)
                n_kept = surv.shape[0]
                assert n_kept >= 1 and len(tree[key]) == n_kept
                kept[:n_kept] = surv
            rows.append((parent, acts, n_kept, kept, 1 if last else 0))
        if not attempts:
            assert not new_keys                                             # a terminal leaf or the horizon: no expansion
    return rows


def main():
    sys.dont_write_bytecode = True
    os.chdir(tempfile.mkdtemp(prefix="golden_tree_"))
    sys.path.insert(0, REF)
    import torch

    import act  # noqa: E402  (reference)
    import datasets  # noqa: E402  (reference)
    import utils  # noqa: E402  (reference)

    out = {}
    strassen, _ = datasets.get_strassen_tensor("cpu")
    rng = np.random.default_rng(7)
    cases = [
        ("S4_T1", 4, 1, 8, 160, 6, 11, strassen.numpy().astype(np.int8)),
        ("S4_T2", 4, 2, 5, 120, 5, 12, strassen.numpy().astype(np.int8)),
        ("S9_T1", 9, 1, 6, 60, 4, 13, utils.build_matmul_tensor(1, 3, 3, 3)[0].numpy().astype(np.int8)),
        ("S4_rand", 4, 1, 8, 120, 6, 14, rng.integers(-1, 2, size=(4, 4, 4)).astype(np.int8)),
        # few distinct actions, deep search: the same states are reached along many paths (transpositions), so the
        # membership test drops many candidates
        ("S4_transp", 4, 1, 6, 300, 8, 15, strassen.numpy().astype(np.int8), 3, 0.1, True),
        ("S9_transp", 9, 1, 6, 150, 8, 16, utils.build_matmul_tensor(1, 3, 3, 3)[0].numpy().astype(np.int8), 3, 0.1, True),
    ]
    for name, S, T, k, n_sim, max_actions, seed, start, *extra in cases:
        rows = record(torch, act, utils, S, T, k, n_sim, max_actions, seed, start, *extra)
        out[f"{name}_parent"] = np.stack([r[0] for r in rows])
        out[f"{name}_actions"] = np.stack([r[1] for r in rows])
        out[f"{name}_n_kept"] = np.array([r[2] for r in rows], np.int32)
        out[f"{name}_kept"] = np.stack([r[3] for r in rows])
        out[f"{name}_commit"] = np.array([r[4] for r in rows], np.uint8)
        nk = out[f"{name}_n_kept"]
        print(f"{name}: {len(rows)} attempts, {int((nk == 0).sum())} with no survivor, "
              f"{int((nk < k).sum())} with at least one candidate dropped, mean kept {nk.mean():.2f} of {k}")
    np.savez_compressed(OUT / "tree_filter.npz", **out)
    print(f"wrote tree_filter.npz ({(OUT / 'tree_filter.npz').stat().st_size / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
