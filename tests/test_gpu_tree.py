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
def test_tree_filter_matches_reference_extend_tree(name, fused):  # noqa: C901
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


@pytest.mark.parametrize("S,B,k", [(16, 1, 1), (16, 5, 8), (16, 131, 3), (16, 37, 70), (16, 2048, 8),
                                   (25, 1, 1), (25, 3, 8), (25, 37, 3), (25, 5, 70), (25, 300, 8)])
def test_expand_keys_fused_at_s16_and_s25(S, B, k):
    """Round 4: at S=16 and S=25 tg_expand_keyed_i8 forms the keys inside the expansion kernel (packed_kernel<.., EXPAND, *,
    true>: a wavefront per parent and shuffles at S=16, a workgroup per parent and per-wavefront partial sums in LDS at
    S=25, whose last chunk holds nine bytes): ragged batches, more actions than one staging tile, null actions, overflowing
    children, and parents whose factors are beyond the packed form (the workgroup's exact fallback writes the children,
    then keys them from memory) -- keys == tg_hash_u64 of the children == the oracle's key, children / done / changed ==
    the oracle."""
    rng = np.random.default_rng(B * 100 + k)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, k, 3 * S)).astype(np.int8)
    ac[::3, 0, :S] = 1                                                  # null actions (u = 0)
    if B > 4:
        st[2] = 127                                                     # children of game 2 overflow
        ac[4, k - 1] = rng.integers(-128, 128, size=3 * S)              # factors beyond the packed form: exact fallback of its workgroup
    kids_o, done_o, changed_o, ovf_o = O.expand_i8(st, ac)
    want = O.state_hash(kids_o.reshape(B * k, S, S, S)).reshape(B, k)
    ovf = torch.zeros((B, k), dtype=torch.uint8, device=DEV)
    kids, done, changed, keys = ops.expand(padded(st), dev(ac), want_keys=True, overflow=ovf)
    assert np.array_equal(host(kids), kids_o) and np.array_equal(host(done), done_o) and np.array_equal(host(changed), changed_o)
    assert np.array_equal(host(ovf), ovf_o)
    assert np.array_equal(host(keys).view(np.uint64), want)
    assert np.array_equal(host(ops.state_hash(kids.flatten(0, 1))).view(np.uint64), want.reshape(-1))


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


# ------------------------------------------------------------------ N1 fused: step + model input in one call
@pytest.mark.parametrize("S,T", [(4, 3), (9, 2)])
def test_step_emit_matches_reference_rollouts(golden, S, T):
    """The 5-step get_child_states rollouts recorded from the reference (T-frame history, act.py:271-274): after every
    fused step the emitted model input equals the reference's (B,T,S,S,S) state, in all three float types, and the ring
    holds the same frames as step() + model_input() would."""
    from mat_mul_amd import TensorGameEnv

    g = golden("next_rows")
    states, actions = g[f"hist_S{S}_T{T}_states"], g[f"hist_S{S}_T{T}_actions"]
    B = states.shape[1]
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        env = TensorGameEnv(B, S, DEV, dim_t=T)
        env.reset(dev(states[0][:, 0]))
        ref = TensorGameEnv(B, S, DEV, dim_t=T)
        ref.reset(dev(states[0][:, 0]))
        for k in range(actions.shape[0]):
            x, sc, done = env.step_observe(dev(actions[k]), dtype=dt)
            assert x.dtype == dt and np.array_equal(x.float().cpu().numpy(), states[k + 1].astype(np.float32)), (S, T, k, dt)
            assert bool((sc == k + 1).all())
            ref.step(dev(actions[k]))
            xr, scr = ref.model_input(dt)
            assert torch.equal(x, xr) and torch.equal(sc, scr) and torch.equal(done, ref.done)
            assert torch.equal(env.state, ref.state) and env.head == ref.head


def test_step_emit_random_against_oracle():
    rng = np.random.default_rng(41)
    for S, B, T in [(4, 1, 1), (4, 3, 2), (4, 70, 4), (4, 257, 5), (4, 65, 1), (16, 9, 3), (25, 3, 2), (5, 4, 2)]:
        frames = rng.integers(-3, 4, size=(B, T, S, S, S)).astype(np.int8)
        ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
        ac[::4] = rng.integers(-60, 60, size=ac[::4].shape)                   # wide factors: the 32-bit redo, overflow
        for head in range(T):
            ring = ops.alloc_ring(B, S, T, DEV)
            ring.copy_(dev(frames))
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            x, sc, done, nxt = ops.step_emit(ring, head, dev(ac), 7.0, overflow=ovf)
            new, want_done, want_ovf = O.step_i8(frames[:, head], ac)
            want = frames.copy()
            want[:, (head + 1) % T] = new
            assert nxt == (head + 1) % T and np.array_equal(host(ring), want), (S, B, T, head)
            order = [(nxt - f) % T for f in range(T)]
            assert np.array_equal(host(x), want[:, order].astype(np.float32)), (S, B, T, head)
            assert np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf) and bool((sc == 7.0).all())
    # shift 2 (StrassenDemoDataset tokens) and guard bytes around the output
    B, S, T = 33, 4, 3
    frames = rng.integers(-2, 3, size=(B, T, S, S, S)).astype(np.int8)
    ac = rng.integers(1, 4, size=(B, 12)).astype(np.int8)
    ring = ops.alloc_ring(B, S, T, DEV)
    ring.copy_(dev(frames))
    buf = torch.full((B * T * 64 + 64,), 99.0, dtype=torch.float16, device=DEV)
    x = buf[32:32 + B * T * 64].view(B, T, S, S, S)
    ops.step_emit(ring, 0, dev(ac), dtype=torch.float16, out=x, shift=2)
    new, _, _ = O.step_i8(frames[:, 0], ac, shift=2)
    assert np.array_equal(host(x[:, 0]).astype(np.int8), new) and bool((buf[:32] == 99).all()) and bool((buf[-32:] == 99).all())


@pytest.mark.parametrize("B,T", [(1, 1), (5, 2), (131, 4), (66, 3)])
def test_step_emit_fused_at_s16(B, T):
    """Round 4: at S=16 tg_step_emit is ONE kernel while the output stays in the caches (s16_step_emit_kernel): every head
    slot of the ring, float32 / float16 / bfloat16, ragged batches (the last workgroup's dead wavefronts), wide factors (the
    32-bit redo, overflow), terminal games, guard elements around the output -- ring, model input, done, overflow and
    scalars against the oracle, and equal to the two-launch path (step, then emit_frames)."""
    S = 16
    rng = np.random.default_rng(B * 10 + T)
    frames = rng.integers(-3, 4, size=(B, T, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
    ac[::4] = rng.integers(-60, 60, size=ac[::4].shape)
    for head in range(T):
        if B > 2:
            frames[2, head] = O.action_to_tensor(ac[2:3])[0].astype(np.int8)     # game 2 is done after the step
        new, want_done, want_ovf = O.step_i8(frames[:, head], ac)
        want = frames.copy()
        want[:, (head + 1) % T] = new
        order = [((head + 1) % T - f) % T for f in range(T)]
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            ring = ops.alloc_ring(B, S, T, DEV)
            ring.copy_(dev(frames))
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            n = B * T * S ** 3
            buf = torch.full((n + 128,), 99.0, dtype=dt, device=DEV)
            x = buf[64:64 + n].view(B, T, S, S, S)
            x, sc, done, nxt = ops.step_emit(ring, head, dev(ac), 3.0, dtype=dt, out=x, overflow=ovf)
            assert nxt == (head + 1) % T and np.array_equal(host(ring), want), (B, T, head, dt)
            assert np.array_equal(x.float().cpu().numpy(), want[:, order].astype(np.float32)), (B, T, head, dt)
            assert np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf) and bool((sc == 3.0).all())
            assert bool((buf[:64] == 99).all()) and bool((buf[-64:] == 99).all())
            # the two-launch path on the same inputs
            ring2 = ops.alloc_ring(B, S, T, DEV)
            ring2.copy_(dev(frames))
            d2 = torch.zeros(B, dtype=torch.uint8, device=DEV)
            ops.step(ring2[:, head], dev(ac), out=ring2[:, (head + 1) % T], done=d2)
            x2, sc2 = ops.emit_frames(ring2, (head + 1) % T, 3.0, dtype=dt)
            assert torch.equal(x, x2) and torch.equal(ring, ring2) and torch.equal(done, d2)
    if B > 2:
        assert want_done[2] == 1


def test_functional_expand_new_candidates_batched():
    """The reference-named wrapper over a BATCH of leaves against the oracle's per-leaf filter sharing one tree."""
    from mat_mul_amd import functional as F

    rng = np.random.default_rng(77)
    B, S, k, T = 40, 4, 6, 2
    pool = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(5, 3 * S)).astype(np.int8)
    state = np.zeros((B, T, S, S, S), np.int8)
    state[:, 0] = rng.integers(-1, 2, size=(B // 4, S, S, S)).astype(np.int8).repeat(4, axis=0)   # repeated leaves
    acts = pool[rng.integers(0, 5, size=(B, k))]
    tt = TranspositionTable(1 << 10, DEV)
    table = set()
    seeds = O.step_i8(state[:8, 0], acts[:8, 0])[0]                                # some children are already tree keys
    tt.insert(ops.state_hash(padded(seeds)))
    O.seen_u64(O.state_hash(seeds), table, insert=True)
    kids, keep, keys, done = F.expand_new_candidates(dev(state), dev(acts), tt)
    want = np.stack([O.tree_filter(state[b, 0], acts[b], table)[0] for b in range(B)])
    assert np.array_equal(host(keep), want) and 0 < int(want.sum()) < want.size
    assert np.array_equal(host(F.state_to_key(dev(state))).view(np.uint64), O.state_hash(state[:, 0]))
    tt.insert(F.state_to_key(dev(state)))
    assert tt.count() == len(table | set(int(x) for x in O.state_hash(state[:, 0])))
