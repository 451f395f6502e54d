"""GPU parity of SURVEY N2's second half: the transposition-table filter of MCTS expansion (act.py:183-195, 209-211).

`tg_seen_u64` against the oracle's Python set, and the whole filter -- tg_expand_i8 -> keys -> tg_seen_u64 -- against
what the reference's own `extend_tree` kept (tests/golden/tree_filter.npz, recorded by make_golden_tree.py)."""
import numpy as np
import pytest
import torch

from mat_mul_amd import TranspositionTable, ops
from oracle import tensor_game as O
from test_oracle_golden import TREE_CASES, replay_tree_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def padded(states_np):
    B, S = states_np.shape[0], states_np.shape[1]
    t = ops.alloc_states(B, S, DEV)
    t.copy_(torch.from_numpy(np.ascontiguousarray(states_np)))
    return t


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", TREE_CASES)
def test_tree_filter_matches_reference_extend_tree(name, fused):
    """Every expansion attempt the reference made, replayed through the C ABI: the survivors must be the reference's
    `not_dupl_actions`, in order, and attempts the reference had to repeat must leave no survivor."""
    tt = TranspositionTable(1 << 12, DEV)

    def filt(parent, acts):
        p = padded(parent[None])
        if fused:
            kids, done, changed, keys = ops.expand(p, dev(acts[None]), want_keys=True)
        else:
            kids, done, changed = ops.expand(p, dev(acts[None]))
            keys = ops.state_hash(kids[0]).unsqueeze(0)
        fresh = tt.fresh(keys, mask=changed)
        want_kept, want_keys, want_changed = O.tree_filter(parent, acts, oracle_table)
        assert np.array_equal(host(keys)[0].view(np.uint64), want_keys)
        assert np.array_equal(host(changed)[0], want_changed) and np.array_equal(host(fresh)[0], want_kept)
        return host(fresh)[0], host(changed)[0]

    oracle_table = set()

    def commit(parent):
        key = ops.state_hash(padded(parent[None]))
        tt.insert(key)
        O.seen_u64(O.state_hash(parent[None]), oracle_table, insert=True)

    replay_tree_fixture(name, filt, commit)
    assert tt.full() is False and len(oracle_table) == tt.count()


def test_seen_against_python_set():
    rng = np.random.default_rng(5)
    table = ops.alloc_seen_table(1 << 14, DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ref = set()
    pool = rng.integers(0, 2 ** 63, size=6000, dtype=np.int64).astype(np.uint64)
    pool[:7] = [0, 1, (1 << 14) + 1, (1 << 15) + 1, 2 ** 64 - 1, 0x9E3779B97F4A7C15, 1 << 14]   # zero key, one probe chain
    for it in range(12):
        n = int(rng.integers(1, 900))
        keys = pool[rng.integers(0, len(pool) if it > 2 else 40, size=n)]        # many repeats, also inside a call
        mask = (rng.random(n) < 0.8).astype(np.uint8) if it % 3 else None
        insert = it % 4 != 3
        got = ops.seen(dev(keys.view(np.int64)), table, mask=None if mask is None else dev(mask), insert=insert, status=status)
        want = O.seen_u64(np.where(keys == 0, np.uint64(0x9E3779B97F4A7C15), keys), ref, mask=mask, insert=insert)
        assert np.array_equal(host(got), want), it
    assert int(status[0]) == 0
    stored = host(table).view(np.uint64)
    assert set(int(x) for x in stored[stored != 0]) == ref and len(ref) == int((stored != 0).sum())
    # 2-D keys (the (B,k) children of an expansion) and an empty call
    k2 = dev(pool[:12].view(np.int64).reshape(3, 4))
    assert tuple(ops.seen(k2, table).shape) == (3, 4)
    assert ops.seen(dev(np.zeros(0, np.int64)), table).numel() == 0


def test_seen_full_table_sets_status_and_loses_nothing_silently():
    table = ops.alloc_seen_table(8, DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    keys = dev(np.arange(1, 13, dtype=np.int64) * 7919)
    ops.seen(keys, table, insert=True, status=status)
    assert int(status[0]) == 1                                   # 12 distinct keys cannot enter 8 slots
    stored = host(table)
    assert (stored != 0).all() and set(stored.tolist()) <= set(host(keys).tolist())
    fresh = host(ops.seen(keys, table))                         # the stored ones are found, the lost ones are fresh
    assert int((fresh == 0).sum()) == 8


def test_expand_keys_match_state_hash():
    """want_keys of expand == state_hash of the children it wrote, every kernel family, packed and byte-offset layouts."""
    rng = np.random.default_rng(17)
    for S, B, k in [(4, 70, 8), (4, 3, 70), (9, 19, 5), (16, 9, 4), (25, 3, 3), (5, 4, 3), (4, 1, 1)]:
        st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
        ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, k, 3 * S)).astype(np.int8)
        for parent in (padded(st), dev(st)):
            kids, done, changed, keys = ops.expand(parent, dev(ac), want_keys=True)
            want = O.state_hash(host(kids).reshape(B * k, S, S, S)).reshape(B, k)
            assert np.array_equal(host(keys).view(np.uint64), want), (S, B, k)
            kids_o, done_o, changed_o, _ = O.expand_i8(st, ac)
            assert np.array_equal(host(kids), kids_o) and np.array_equal(host(done), done_o)
