"""Pin the oracle (oracle/tensor_game.py) against the reference's own outputs.

The fixtures under tests/golden/ were produced by RUNNING the reference
(tests/golden/make_golden.py).  Everything here is CPU-only."""
import hashlib
from pathlib import Path

import numpy as np
import pytest

from oracle import tensor_game as O


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int8).tobytes()).hexdigest()[:16]


# values captured from the reference import
STRASSEN_SHA = ["dd3016dbee83d297", "487bdd231b5807db", "1b6c8c90086b2432", "29c27b2476edbd9d",
                "1b72f1b67ec9cee7", "136ac19825a1a269", "589b154d716d4360", "f5a5fd42d16a2030"]
STRASSEN_NNZ = [8, 12, 12, 12, 10, 8, 4, 0]


def test_strassen_replay(golden):
    g = golden("strassen")
    assert [sha16(s) for s in g["replay"]] == STRASSEN_SHA
    # notebook strassen_example.ipynb cell 4 (printed token table, shift=1)
    assert g["tokens"][0].tolist() == [2, 1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 2]
    assert g["tokens"][6].tolist() == [1, 2, 1, 0, 1, 1, 2, 2, 2, 1, 1, 1]
    tensor, tokens = O.uvw_to_demo(g["uu"], g["vv"], g["ww"], shift=1)
    assert np.array_equal(tensor, g["tensor"]) and np.array_equal(tokens, g["tokens"])
    assert np.array_equal(tensor, g["matmul_2"])
    assert np.array_equal(O.build_matmul_tensor(1, 2, 2, 2)[0], g["matmul_2"])
    state = g["tensor"][None].astype(np.int8)
    for k in range(7):
        state, done, ovf = O.step_i8(state, g["tokens"][k][None], shift=1)
        assert np.array_equal(state[0], g["replay"][k + 1])
        assert int(done[0]) == int(g["done"][k + 1]) == int(k == 6)
        assert int(ovf[0]) == 0
        assert int(O.nnz_per_game(state)[0]) == STRASSEN_NNZ[k + 1]
    # whole replay in one call
    fin, done_step, ovf = O.step_many_i8(g["tensor"][None].astype(np.int8), g["tokens"][None], shift=1)
    assert not fin.any() and done_step.tolist() == [6] and ovf.tolist() == [0]


def test_strassen_dataset_448(golden):
    g = golden("strassen")
    assert g["ds_states"].shape == (448, 4, 4, 4)
    assert sha16(g["ds_states"][:, None]) == "46403915015e57d5"
    assert set(np.unique(g["ds_actions"]).tolist()) <= {1, 2, 3}         # shift 2 (datasets.py:397)
    new, done, ovf = O.step_i8(g["ds_states"], g["ds_actions"], shift=2)
    last = g["ds_rewards"] == -1
    assert int(last.sum()) == 7
    assert np.array_equal(done.astype(bool), last)                      # exactly the last-move pairs terminate
    assert not ovf.any()
    # every state is strassen_tensor minus a subset of the 7 terms: stepping never leaves [-2,2]
    assert np.abs(new).max() <= 2


def test_build_matmul_tensor(golden):
    g = golden("matmul_tensors")
    for n in (2, 3, 4, 5):
        for T in (1, 2):
            ours = O.build_matmul_tensor(T, n, n, n)
            assert np.array_equal(ours, g[f"n{n}_t{T}"])
            assert int(ours.sum()) == n ** 3
        assert np.array_equal(O.reset_matmul_i8(3, n)[2], g[f"n{n}_t1"][0])
    # canonical <n,n,n>: T[a*n+j, j*n+c, a*n+c] = 1
    n = 3
    t = np.zeros((9, 9, 9), np.int64)
    for a in range(n):
        for j in range(n):
            for c in range(n):
                t[a * n + j, j * n + c, a * n + c] = 1
    assert np.array_equal(O.build_matmul_tensor(1, n, n, n)[0], t)
    with pytest.raises(ValueError):
        O.build_matmul_tensor(1, 2, 3, 4)


def _step_tags(g):
    return sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_state")})


def test_get_child_states_cases(golden):
    g = golden("step_cases")
    tags = _step_tags(g)
    assert len(tags) == 12
    for tag in tags:
        st, ac = g[tag + "_state"], g[tag + "_actions"]
        kids = O.get_child_states(st, ac, shift=1)
        kids = np.stack(kids, axis=1)
        assert np.array_equal(kids, g[tag + "_children"]), tag
        B, k = ac.shape[:2]
        done = np.stack([O.done_per_game(kids[:, i, 0]) for i in range(k)], axis=1)
        assert np.array_equal(done.astype(np.uint8), g[tag + "_done"]), tag
        assert g[tag + "_done"].any() or B == 1 or True
        # verbatim game-0-only tensor_factorized on the batched child state
        tf = [O.tensor_factorized(kids[:, i]) for i in range(k)]
        assert np.array_equal(np.array(tf, np.uint8), g[tag + "_tf_verbatim"]), tag
        nn = O.remove_null_actions(st, [kids[:, i] for i in range(k)])
        assert np.array_equal(np.array(nn, np.int64), g[tag + "_nonnull_batch"]), tag
        # build semantics == reference on the head frame
        kids8, done8, changed8, ovf8 = O.expand_i8(st[:, 0], ac, shift=1)
        assert np.array_equal(kids8, g[tag + "_children"][:, :, 0]), tag
        assert np.array_equal(done8, g[tag + "_done"]), tag
        assert np.array_equal(changed8, g[tag + "_changed"]), tag
        assert not ovf8.any()
        new8, d8, o8 = O.step_i8(st[:, 0], ac[:, 0], shift=1)
        assert np.array_equal(new8, g[tag + "_children"][:, 0, 0]) and np.array_equal(d8, g[tag + "_done"][:, 0])
    assert sum(int(g[t + "_done"].sum()) for t in tags) > 50      # the terminal branch is exercised


def test_action_to_tensor(golden):
    g = golden("action_to_tensor")
    for S in (4, 9, 16, 25):
        ac = g[f"S{S}_actions"]
        assert np.array_equal(O.action_to_tensor(ac), g[f"S{S}_batched"])
        assert np.array_equal(np.stack([O.action_to_tensor(a) for a in ac]), g[f"S{S}_single"])
        assert np.array_equal(O.uvw_to_tensor(O.action_to_uvw(ac, shift=2)), g[f"S{S}_shift2"])
    assert np.array_equal(O.action_to_tensor(g["wide_actions"]), g["wide_tensor"])


def test_synthetic_demos_deterministic_half(golden):
    g = golden("synthetic_demos")
    names = sorted(k[: -len("_tokens")] for k in g.files if k.endswith("_tokens") and "_item" not in k)
    assert len(names) >= 15
    for nm in names:
        tok, tgt = g[nm + "_tokens"], g[nm + "_target"]
        ours, ovf = O.gen_from_factors_i8(tok[None], shift=1)
        assert np.array_equal(ours[0], tgt) and not ovf.any(), nm
        # each accepted term is non-null (utils.py:229)
        assert all((O.action_to_tensor(a) != 0).any() for a in tok)
        # replaying all actions returns to zero at the last step (pure sum, order independent)
        fin, done_step, _ = O.step_many_i8(tgt[None], tok[None], shift=1)
        assert not fin.any() and 0 <= int(done_step[0]) <= len(tok) - 1
        fin2, _, _ = O.step_many_i8(tgt[None], tok[None, ::-1], shift=1)
        assert not fin2.any()


def test_take_actions_and_getitem(golden):
    g = golden("synthetic_demos")
    names = sorted(k[: -len("_suffix_states")] for k in g.files if k.endswith("_suffix_states"))
    assert names
    for nm in names:
        tok, tgt, suf = g[nm + "_tokens"], g[nm + "_target"], g[nm + "_suffix_states"]
        R = len(tok)
        T = int(nm.split("_T")[1].split("_")[0])
        for i in range(R):
            assert np.array_equal(O.take_actions(list(tok[i + 1:]), tgt), suf[i]), nm
            frames, scalar, action, reward = O.demo_getitem(list(tok), tgt, i, T)
            assert np.array_equal(frames, g[f"{nm}_item{i}_frames"]), (nm, i)
            assert [scalar, reward] == g[f"{nm}_item{i}_meta"].tolist()
            assert np.array_equal(action, g[f"{nm}_item{i}_action"])


def test_ref_dtype_port(golden):
    """oracle/ref_dtype_torch.py (the cpu_baseline 'port') == the reference's outputs."""
    import torch
    from oracle import ref_dtype_torch as P
    g = golden("step_cases")
    for tag in _step_tags(g):
        st = torch.from_numpy(g[tag + "_state"].astype(np.float32))
        ac = torch.from_numpy(g[tag + "_actions"].astype(np.int64))
        kids = P.get_child_states(st, ac)
        assert kids[0].dtype == torch.float32
        got = np.stack([c.numpy() for c in kids], axis=1)
        assert np.array_equal(got, g[tag + "_children"].astype(np.float32)), tag
        new, done = P.env_step(st, ac[:, :1])
        assert np.array_equal(done.numpy().astype(np.uint8), g[tag + "_done"][:, 0]), tag


def test_next_rows_oracle(golden):
    """N1-N3: history rollouts, exact slice rank vs the reference's get_rank, hash sanity."""
    g = golden("next_rows")
    for S in (4, 9, 16, 25):
        assert np.array_equal(O.slice_rank_exact(g[f"rank_S{S}_state"]), g[f"rank_S{S}_rank"]), S
    for tag in ("hist_S4_T3", "hist_S9_T2"):
        states, acts = g[tag + "_states"], g[tag + "_actions"]
        cur = states[0]
        for k in range(len(acts)):
            cur = O.get_child_states(cur, acts[k][:, None, :])[0]
            assert np.array_equal(cur, states[k + 1]), (tag, k)
            x, sc = O.model_input(cur, k + 1)
            assert x.dtype == np.float32 and sc.shape == (cur.shape[0], 1) and sc[0, 0] == k + 1
    s = golden("strassen")["ds_states"]
    h = O.state_hash(s)
    strings = ["_".join(map(str, x.reshape(-1))) for x in s]      # the reference's key (utils.py:164-169)
    assert len(set(h.tolist())) == len(set(strings))              # equal keys <=> equal states on all 448
    assert len({(a, b) for a, b in zip(h.tolist(), strings)}) == len(set(strings))


def test_notebook_known_answers(golden):
    """notebooks/strassen_example.ipynb: the printed Strassen tensor (cell 5), the token table (cell 4),
    the 448-pair count (cell 8) and the outer-product == einsum check (cells 14-17)."""
    g = golden("strassen")
    t = g["tensor"]
    ones = {(0, 0, 0), (0, 1, 1), (1, 2, 0), (1, 3, 1), (2, 0, 2), (2, 1, 3), (3, 2, 2), (3, 3, 3)}   # cell 5 printout
    assert {tuple(ix) for ix in np.argwhere(t == 1)} == ones and int(np.abs(t).sum()) == 8
    assert g["tokens"].tolist() == [[2, 1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 2], [1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 2, 0],
                                    [2, 1, 1, 1, 1, 2, 1, 0, 1, 2, 1, 2], [1, 1, 1, 2, 0, 1, 2, 1, 2, 1, 2, 1],
                                    [2, 2, 1, 1, 1, 1, 1, 2, 0, 2, 1, 1], [0, 1, 2, 1, 2, 2, 1, 1, 1, 1, 1, 2],
                                    [1, 2, 1, 0, 1, 1, 2, 2, 2, 1, 1, 1]]                                 # cell 4 printout
    assert len(g["ds_states"]) == 448                                                                     # cell 8
    for u, v, w in zip(g["uu"], g["vv"], g["ww"]):                                                        # cells 14-17
        assert np.array_equal(O.uvw_to_tensor((u, v, w)), np.einsum("p,qr->pqr", u, np.outer(v, w)))
    first = O.uvw_to_tensor((g["uu"][0], g["vv"][0], g["ww"][0]))                                          # cell 14 printout
    assert {tuple(ix) for ix in np.argwhere(first == 1)} == {(0, 0, 0), (0, 0, 3), (0, 3, 0), (0, 3, 3),
                                                             (3, 0, 0), (3, 0, 3), (3, 3, 0), (3, 3, 3)}


# ---------------------------------------------------------------------------------------------
# N2, second half: the tree filter of extend_tree (act.py:183-195, 209-211), recorded from the reference
# ---------------------------------------------------------------------------------------------
TREE_CASES = ["S4_T1", "S4_T2", "S9_T1", "S4_rand", "S4_transp", "S9_transp"]


def replay_tree_fixture(name, filter_fn, commit_fn):
    """Walks the recorded expansion attempts in order.  filter_fn(parent, actions) -> kept mask (k,);
    commit_fn(parent) records the expanded state.  Asserts the survivors are the reference's, in order."""
    g = np.load(Path(__file__).resolve().parent / "golden" / "tree_filter.npz")
    parents, actions, n_kept, kept, commit = (g[f"{name}_{f}"] for f in ("parent", "actions", "n_kept", "kept", "commit"))
    dropped_by_tree = 0
    for e in range(len(parents)):
        mask, changed = filter_fn(parents[e], actions[e])
        mask = np.asarray(mask).astype(bool)
        assert int(mask.sum()) == int(n_kept[e]), (name, e)
        assert np.array_equal(actions[e][mask], kept[e][: n_kept[e]]), (name, e)
        dropped_by_tree += int((np.asarray(changed).astype(bool) & ~mask).sum())
        if commit[e]:
            assert n_kept[e] >= 1
            commit_fn(parents[e])
        else:
            assert n_kept[e] == 0      # the reference asked the network again
    return dropped_by_tree


@pytest.mark.parametrize("name", TREE_CASES)
def test_tree_filter_matches_reference_extend_tree(name):
    table = set()

    def filt(parent, acts):
        kept, _, changed = O.tree_filter(parent, acts, table)
        return kept, changed

    def commit(parent):
        O.seen_u64(O.state_hash(parent[None]), table, insert=True)

    dropped = replay_tree_fixture(name, filt, commit)
    if name.endswith("transp"):
        assert dropped >= 20, "the transposition cases must exercise the membership test, not only null actions"


def test_seen_semantics():
    t = set()
    k = np.array([5, 7, 5, 0, 9], np.uint64)
    assert O.seen_u64(k, t, insert=True).tolist() == [1, 1, 1, 1, 1]     # equal keys of one call are all fresh
    assert t == {5, 7, 0, 9}
    assert O.seen_u64(k, t, mask=np.array([1, 0, 1, 1, 1])).tolist() == [0, 0, 0, 0, 0]
    assert O.seen_u64(np.array([11, 5], np.uint64), t, mask=np.array([0, 1]), insert=True).tolist() == [0, 0]
    assert 11 not in t                                                    # masked-out keys are not recorded
