"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Bit-exact everywhere (integer/byte work).  Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

import mat_mul_amd
from mat_mul_amd import TensorGameEnv, SyntheticDemos, functional as F, ops
from oracle import tensor_game as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def padded(states_np):
    """int8 (B,S,S,S) numpy -> device tensor with a 16-byte-multiple game stride (fast path)."""
    B, S = states_np.shape[0], states_np.shape[1]
    t = ops.alloc_states(B, S, DEV)
    t.copy_(torch.from_numpy(np.ascontiguousarray(states_np)))
    return t


def rand_case(rng, B, S, k=None, lo=-2, hi=3, terminal_every=5):
    st = rng.integers(lo, hi, size=(B, S, S, S)).astype(np.int8)
    shape = (B, 3 * S) if k is None else (B, k, 3 * S)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=shape).astype(np.int8)
    first = ac if k is None else ac[:, 0]
    for b in range(0, B, terminal_every):  # terminal games: state == action tensor
        st[b] = O.action_to_tensor(first[b]).astype(np.int8)
    return st, ac


@pytest.fixture(scope="module", autouse=True)
def _native_loaded():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    assert mat_mul_amd._lib.lib.tg_abi_version() == mat_mul_amd._lib.TG_ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libtensorgame.so" in maps, "the HIP library is not the one loaded"


# ------------------------------------------------------------------ golden fixtures via the C ABI
def test_strassen_replay_golden(golden):
    g = golden("strassen")
    env = TensorGameEnv(1, 4, DEV)
    state = env.reset()
    assert np.array_equal(host(state)[0], g["replay"][0])            # reset == build_matmul_tensor(1,2,2,2)
    for k in range(7):
        state, done = env.step(dev(g["tokens"][k][None]))
        assert np.array_equal(host(state)[0], g["replay"][k + 1])
        assert int(done[0]) == int(g["done"][k + 1])
        assert int(env.nnz()[0]) == [12, 12, 12, 10, 8, 4, 0][k]
    assert not env.any_overflow()
    env.reset()
    state, done_step = env.step_many(dev(g["tokens"][None]))
    assert not host(state).any() and host(done_step).tolist() == [6]
    # reference-named functions
    tensor, tokens = F.uvw_to_demo(dev(g["uu"]), dev(g["vv"]), dev(g["ww"]), DEV)
    assert np.array_equal(host(tensor), g["tensor"]) and np.array_equal(host(tokens), g["tokens"])
    assert bool(F.tensor_factorized(state.unsqueeze(1)))
    out = F.take_actions(list(dev(g["tokens"])), dev(g["tensor"]))
    assert not host(out).any()


@pytest.mark.parametrize("shift", [0, 1, 2, 3, 4, -1])
def test_s4_digit_form_boundaries(shift):
    """tg_step_i8 at S=4 takes the digit form (one 32-bit multiply-add per row of a slice) only where no byte can
    carry into its neighbour: every token <= 3, 0 <= shift <= 3, and the slice's L1 norm <= 127 - F^3.  Games placed
    on both sides of each of those limits -- and slices full of -128 / 127, where a carry WOULD cross -- must equal
    the oracle bit for bit, flags included."""
    rng = np.random.default_rng(100 + shift)
    F3 = max(shift, 3 - shift) ** 3 if 0 <= shift <= 3 else 27
    limit = 127 - F3
    B = 64 * 6 + 5
    st = np.zeros((B, 4, 4, 4), dtype=np.int8)
    ac = rng.integers(0, 4, size=(B, 12)).astype(np.int8)                  # all tokens <= 3
    for b in range(B):
        kind = b % 8
        for i in range(4):
            # one big entry per slice puts the slice norm at limit-1, limit, limit+1, 127, 128 (as -128) ...
            big = [limit - 1, limit, limit + 1, 127, -128, -limit, -(limit + 1), 5][kind]
            big = int(np.clip(big, -128, 127))
            st[b, i].flat[rng.integers(0, 16)] = big
            if kind in (0, 5):                                               # spread the rest of the norm thinly
                pass
            elif kind == 7:                                                  # many small entries, norm around the limit
                st[b, i] = rng.integers(-8, 9, size=(4, 4))
    ac[3::11, rng.integers(0, 12)] = 4                                       # a token beyond the digit form's vocabulary
    ac[5::13] = rng.integers(-128, 128, size=ac[5::13].shape)               # wide tokens (negative bytes too)
    st[7::17] = rng.choice([-128, 127], size=st[7::17].shape)               # every byte at the edge
    want, want_done, want_ovf = O.step_i8(st, ac, shift=shift)
    for inplace in (False, True):
        t = padded(st)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, done = ops.step(t, dev(ac), out=t if inplace else None, overflow=ovf, shift=shift)
        assert np.array_equal(host(out), want)
        assert np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf)
    assert want_ovf.any() and not want_ovf.all()


@pytest.mark.parametrize("S,B", [(16, 1), (16, 5), (16, 70), (16, 260), (25, 1), (25, 3), (25, 37), (25, 2051), (4, 70), (9, 19), (5, 6)])
def test_step_tracked_equals_the_full_step_and_keeps_nnz(S, B):
    """tg_step_tracked_i8: the in-place step that loads only the rows the action touches, with the number of non-zero
    entries carried per game.  State, done and overflow must equal tg_step_i8's (hence the oracle's), nnz the count of
    non-zero entries of the new state -- over several steps of one rollout (the count is carried), sparse and dense
    factors (more candidates than a queue batch holds), wide tokens (32-bit redo, wrapped bytes, overflow flag), null
    actions, states at the int8 edge, three shifts.  S = 16 / 25 take the sparse kernels (S = 25 from 2 048 games on; fewer:
    the full step's kernel with the count), every other S the full step + tg_done_i8 inside the call."""
    rng = np.random.default_rng(7 * S + B)
    for case, shift in (("sparse", 1), ("dense", 1), ("wide", 1), ("wide", -2), ("edge", 1), ("null", 2), ("sparse", 3)):
        st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
        if case == "edge":
            st = rng.choice([-128, 127, 0, 1], size=st.shape).astype(np.int8)
        t = padded(st)
        _, nnz = ops.done(t, want_nnz=True)
        assert np.array_equal(host(nnz), np.count_nonzero(st.reshape(B, -1), axis=1))
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        cur, want_ovf = st, np.zeros(B, dtype=np.uint8)
        for k in range(3):
            ac = (rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)) + (shift - 1)).astype(np.int8)
            if case == "dense":
                ac = (rng.integers(0, 3, size=(B, 3 * S)) + (shift - 1)).astype(np.int8)
            if case == "wide":
                ac[::2] = rng.integers(-128, 128, size=ac[::2].shape)
            if case == "null":
                ac[:, :S] = shift                                               # u == 0: nothing changes
            cur, want_done, o = O.step_i8(cur, ac, shift=shift)
            want_ovf |= o
            _, done = ops.step_tracked(t, dev(ac), nnz, overflow=ovf, shift=shift)
            assert np.array_equal(host(t), cur), (S, B, case, shift, k)
            assert np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf), (S, B, case, shift, k)
            assert np.array_equal(host(nnz), np.count_nonzero(cur.reshape(B, -1), axis=1)), (S, B, case, shift, k)


def test_env_track_nnz_matches_the_plain_env():
    """TensorGameEnv(track_nnz=True) steps with the tracked kernel and serves nnz() from the carried count: same states,
    done flags and counts as the plain env over a rollout, also through graph_stepper and after step_many."""
    rng = np.random.default_rng(5)
    for S, B in [(16, 40), (25, 6), (4, 33)]:
        st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
        a, b = TensorGameEnv(B, S, DEV), TensorGameEnv(B, S, DEV, track_nnz=True)
        a.reset(dev(st))
        b.reset(dev(st))
        assert torch.equal(a.nnz(), b.nnz())
        for k in range(4):
            ac = dev(rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8))
            sa, da = a.step(ac)
            sb, db = b.step(ac)
            assert torch.equal(sa, sb) and torch.equal(da, db) and torch.equal(a.nnz(), b.nnz()), (S, k)
        tokens = torch.zeros((B, 3 * S), dtype=torch.int8, device=DEV)
        stepper = b.graph_stepper(tokens)
        for k in range(3):
            ac = dev(rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8))
            tokens.copy_(ac)
            a.step(ac)
            stepper()
            torch.cuda.synchronize()
            assert torch.equal(a.state, b.state) and torch.equal(a.done, b.done) and torch.equal(a.nnz(), b.nnz()), (S, k)
        many = dev(rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3, 3 * S)).astype(np.int8))
        a.step_many(many)
        b.step_many(many)
        assert torch.equal(a.state, b.state) and torch.equal(a.nnz(), b.nnz())
    with pytest.raises(mat_mul_amd.TensorGameError):
        TensorGameEnv(4, 4, DEV, dim_t=2, track_nnz=True)


@pytest.mark.parametrize("S,below,above", [(16, 11999, 12000), (25, 2047, 2048)])
def test_env_picks_the_tracked_step_by_batch_size(S, below, above):
    """TensorGameEnv(track_nnz=None) takes tg_step_tracked_i8 from the measured crossovers on (S=16:
    12 000 games, S=25: 2 048) when dim_t == 1 -- both sides of each crossover roll a demonstration out to zero with the
    same states, done flags and counts as an env forced the other way; dim_t > 1 and S=4 never track."""
    R = 6
    for B in (below, above):
        auto = TensorGameEnv(B, S, DEV)
        assert (auto._nnz is not None) == (B == above)
        forced = TensorGameEnv(B, S, DEV, track_nnz=(B != above))       # the other kernel
        tok, tgt = ops.gen_demos(B, S, R, DEV, seed=B)
        auto.reset(tgt)
        forced.reset(tgt)
        assert torch.equal(auto.nnz(), forced.nnz())
        for k in range(R):
            sa, da = auto.step(tok[:, k].contiguous())
            sf, df = forced.step(tok[:, k].contiguous())
            assert torch.equal(da, df) and torch.equal(auto.nnz(), forced.nnz()), (B, k)
        assert torch.equal(auto.state, forced.state) and not bool(auto.state.any()) and bool(auto.done.all())
        del auto, forced, tok, tgt
        torch.cuda.empty_cache()
    assert TensorGameEnv(above, S, DEV, dim_t=2)._nnz is None and TensorGameEnv(1 << 16, 4, DEV)._nnz is None
    assert TensorGameEnv(above, S, DEV, track_nnz=False)._nnz is None


def test_step_tracked_replays_a_demo_to_zero():
    """BASELINE config 5's shape at a size the oracle is not needed for: generate, then replay the demo's own actions with
    the tracked step -- every game ends all zero with nnz = 0 and done = 1 exactly at the last action."""
    for S, B, R in [(25, 4096, 12), (16, 16384, 10)]:
        tok, tgt = ops.gen_demos(B, S, R, DEV, seed=9)
        _, nnz = ops.done(tgt, want_nnz=True)
        done = torch.zeros(B, dtype=torch.uint8, device=DEV)
        for k in reversed(range(R)):
            ops.step_tracked(tgt, tok[:, k].contiguous(), nnz, done=done)
        assert not bool(tgt.any()) and not bool(nnz.any()) and bool(done.all()), (S, B)


def test_strassen_dataset_448_golden(golden):
    g = golden("strassen")
    st = padded(g["ds_states"])
    new, done = ops.step(st, dev(g["ds_actions"]), shift=2)
    new_o, done_o, _ = O.step_i8(g["ds_states"], g["ds_actions"], shift=2)
    assert np.array_equal(host(new), new_o) and np.array_equal(host(done), done_o)
    assert np.array_equal(host(done).astype(bool), g["ds_rewards"] == -1) and int(done.sum()) == 7


def test_build_matmul_tensor_golden(golden):
    g = golden("matmul_tensors")
    for n in (2, 3, 4, 5):
        for T in (1, 2):
            assert np.array_equal(host(F.build_matmul_tensor(T, n, n, n, DEV)), g[f"n{n}_t{T}"])
        env = TensorGameEnv(37, n * n, DEV)
        s = host(env.reset())
        assert all(np.array_equal(s[b], g[f"n{n}_t1"][0]) for b in (0, 17, 36))
        assert host(env.nnz()).tolist() == [n ** 3] * 37


def test_get_child_states_golden(golden):
    g = golden("step_cases")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_state")})
    assert len(tags) == 12
    for tag in tags:
        st, ac = g[tag + "_state"], g[tag + "_actions"]
        B, k = ac.shape[:2]
        kids = F.get_child_states(dev(st), dev(ac))
        assert len(kids) == k
        got = np.stack([host(c) for c in kids], axis=1)
        assert np.array_equal(got, g[tag + "_children"]), tag
        children, done, changed = ops.expand(padded(st[:, 0]), dev(ac))
        assert np.array_equal(host(children), g[tag + "_children"][:, :, 0]), tag
        assert np.array_equal(host(done), g[tag + "_done"]), tag
        assert np.array_equal(host(changed), g[tag + "_changed"]), tag
        assert F.remove_null_actions(dev(st), kids) == g[tag + "_nonnull_batch"].tolist(), tag
        tf = [bool(F.tensor_factorized(c)) for c in kids]
        assert tf == g[tag + "_tf_verbatim"].astype(bool).tolist(), tag
        new, d1 = ops.step(padded(st[:, 0]), dev(ac[:, 0]))
        assert np.array_equal(host(new), g[tag + "_children"][:, 0, 0]) and np.array_equal(host(d1), g[tag + "_done"][:, 0])


def test_action_to_tensor_golden(golden):
    g = golden("action_to_tensor")
    for S in (4, 9, 16, 25):
        ac = g[f"S{S}_actions"]
        assert np.array_equal(host(F.action_to_tensor(dev(ac))), g[f"S{S}_batched"])
        assert np.array_equal(host(F.action_to_tensor(dev(ac[2]))), g[f"S{S}_single"][2])
        assert np.array_equal(host(F.action_to_tensor(dev(ac), shift=2)), g[f"S{S}_shift2"])
    wide = g["wide_tensor"]
    ovf = torch.zeros(5, dtype=torch.uint8, device=DEV)
    out = ops.gen_from_factors(dev(g["wide_actions"])[:, None], 4, overflow=ovf)
    assert np.array_equal(host(out), wide.astype(np.int8))
    assert np.array_equal(host(ovf).astype(bool), ((wide < -128) | (wide > 127)).reshape(5, -1).any(axis=1))


def test_synthetic_demo_golden(golden):
    g = golden("synthetic_demos")
    names = sorted(k[: -len("_tokens")] for k in g.files if k.endswith("_tokens") and "_item" not in k)
    for nm in names:
        tok, tgt = g[nm + "_tokens"], g[nm + "_target"]
        S = tgt.shape[-1]
        out = ops.gen_from_factors(dev(tok[None]), S)
        assert np.array_equal(host(out)[0], tgt), nm
        fin, done_step = ops.step_many(padded(tgt[None]), dev(tok[None]))
        assert not host(fin).any() and 0 <= int(done_step[0]) < len(tok)
    names = sorted(k[: -len("_suffix_states")] for k in g.files if k.endswith("_suffix_states"))
    for nm in names:
        tok, tgt, suf = g[nm + "_tokens"], g[nm + "_target"], g[nm + "_suffix_states"]
        for i in range(len(tok)):
            assert np.array_equal(host(F.take_actions(dev(tok[i + 1:]), dev(tgt))), suf[i]), (nm, i)


# ------------------------------------------------------------------ seeded random vs the oracle
@pytest.mark.parametrize("S", [1, 2, 3, 4, 5, 8, 9, 16, 25, 32])
@pytest.mark.parametrize("B", [1, 3, 70, 257])
def test_step_matches_oracle(S, B):
    if S >= 25 and B > 70:
        B = 33
    rng = np.random.default_rng(S * 1000 + B)
    st, ac = rand_case(rng, B, S)
    want, want_done, want_ovf = O.step_i8(st, ac)
    for layout in ("padded", "packed", "offset"):
        if layout == "padded":
            t = padded(st)
        elif layout == "packed":
            t = dev(st)
        else:  # games start 3 bytes into a buffer: nothing is aligned -> byte path
            buf = torch.zeros(B * S ** 3 + 3, dtype=torch.int8, device=DEV)
            t = buf[3:].view(B, S, S, S)
            t.copy_(dev(st))
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, done = ops.step(t, dev(ac), overflow=ovf)
        assert np.array_equal(host(out), want), (S, B, layout)
        assert np.array_equal(host(done), want_done), (S, B, layout)
        assert np.array_equal(host(t), st), "input must be untouched out of place"
        assert not host(ovf).any() and not want_ovf.any()
        out2, done2 = ops.step(t, dev(ac), out=t)  # in place
        assert out2.data_ptr() == t.data_ptr()
        assert np.array_equal(host(t), want) and np.array_equal(host(done2), want_done)
    assert want_done.sum() >= 1


@pytest.mark.parametrize("S,B", [(16, 1), (16, 130), (25, 1), (25, 21), (16, 24576), (9, 1), (9, 3), (9, 67), (9, 4099)])
def test_step_direct_kernels_sparse_dense_null_actions(S, B):
    """The direct S=9 / S=16 / S=25 step kernels: sparse actions (S=16: the compacted queue), dense ones (more candidate rows
    than the queue holds: every chunk by its own lane), null actions, finished games; in place and out of place.
    B = 24 576 at S=16 is 96 MiB of states: the variant that stores whole 128-byte lines."""
    rng = np.random.default_rng(S * 31 + B)
    for case in (("sparse", "dense", "null") if B < 5000 else ("sparse",)):
        st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
        ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
        if case == "dense":
            ac = rng.integers(0, 3, size=(B, 3 * S)).astype(np.int8)
        if case == "null":
            ac[:, :S] = 1
            st[::2] = 0
        if B >= 5000:  # a few dense and a few finishing games inside the big batch
            ac[::97] = rng.integers(0, 3, size=ac[::97].shape)
            st[5::101] = O.gen_from_factors_i8(ac[5::101, None, :])[0]
        want, want_done, want_ovf = O.step_i8(st, ac)
        assert not want_ovf.any()
        for inplace in (False, True):
            t = padded(st)
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            out, done = ops.step(t, dev(ac), out=t if inplace else None, overflow=ovf)
            assert np.array_equal(host(out), want), (S, B, case, inplace)
            assert np.array_equal(host(done), want_done) and not host(ovf).any(), (S, B, case, inplace)
            assert inplace or np.array_equal(host(t), st)
        assert case != "null" or want_done[::2].all()
        assert B < 5000 or want_done[5::101].all()


@pytest.mark.parametrize("S", [4, 9, 16, 25, 6])
def test_step_wide_tokens_and_overflow(S):
    rng = np.random.default_rng(77 + S)
    B = 41
    st = rng.integers(-128, 128, size=(B, S, S, S)).astype(np.int8)
    ac = rng.integers(-3, 6, size=(B, 3 * S)).astype(np.int8)
    ac[::2] = rng.integers(0, 3, size=(len(ac[::2]), 3 * S))
    st[::2] = np.clip(st[::2], -100, 100)  # these games cannot overflow for shift=1
    for shift in (1, 2, -1):
        want, want_done, want_ovf = O.step_i8(st, ac, shift=shift)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, done = ops.step(padded(st), dev(ac), overflow=ovf, shift=shift)
        assert np.array_equal(host(out), want) and np.array_equal(host(done), want_done)
        assert np.array_equal(host(ovf), want_ovf) and want_ovf.any()
        assert shift != 1 or not want_ovf[::2].any()
        sticky = torch.ones(B, dtype=torch.uint8, device=DEV)           # never cleared by a call
        ops.step(padded(st), dev(ac), overflow=sticky, shift=shift)
        assert host(sticky).all()


@pytest.mark.parametrize("S", [4, 16, 9, 25, 3])
def test_step_extreme_factors_full_int8_range(S):
    """Tokens over the whole int8 range and shifts up to far beyond it: |u v| can pass 2^15 and
    |u v w| 2^23.  The 16-bit kernels must hand exactly these cases to their 32-bit form: stored
    bytes are the wrapped low bytes of the true result, and the flag is set iff it left int8."""
    rng = np.random.default_rng(4242 + S)
    B = 96
    st = rng.integers(-128, 128, size=(B, S, S, S)).astype(np.int8)
    for shift in (1, 127, -127, 128, 300, -1000):
        ac = rng.integers(-128, 128, size=(B, 3 * S)).astype(np.int8)
        zero_tok = np.int8(shift) if -128 <= shift <= 127 else None
        if zero_tok is not None:
            # plant zero factors: whole w (product vanishes although u v is huge), single entries elsewhere
            ac[0::4, 2 * S:] = zero_tok
            ac[1::4, :S] = np.where(rng.random((len(ac[1::4]), S)) < 0.7, zero_tok, ac[1::4, :S])
            ac[2::4, S:2 * S] = np.where(rng.random((len(ac[2::4]), S)) < 0.7, zero_tok, ac[2::4, S:2 * S])
            ac[3::4] = np.clip(ac[3::4].astype(int), shift - 2, shift + 2).astype(np.int8)  # small factors
        want, want_done, want_ovf = O.step_i8(st, ac, shift=shift)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, done = ops.step(padded(st), dev(ac), overflow=ovf, shift=shift)
        assert np.array_equal(host(out), want), shift
        assert np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf), shift
        if zero_tok is not None:
            assert not want_ovf[0::4].any() and want_ovf.any()


@pytest.mark.parametrize("S,K", [(25, 6), (25, 12), (25, 40), (25, 64), (25, 127), (16, 20), (16, 40), (16, 64), (16, 100), (9, 30), (9, 48), (9, 90)])
def test_step_many_matrix_core_path_verdicts(S, K):
    """tg_mfma.h many_mfma_kernel certifies a game only when its overflow bound holds and the zero state is
    reached at the last step or never; everything else goes to the lattice kernels through done_step.  One
    batch with every kind of game, all bit-exact against the oracle (state, done_step, overflow):
    demo replays (done at K-1), early termination (the tail of the action list cancels), never done,
    start at zero, a step that overflows in the middle but returns, large factors, and games that pass
    through zero and leave again."""
    rng = np.random.default_rng(31 * S + K)
    B = 24
    thr = O.categorical_thresholds((0.15, 0.7, 0.15))
    tok, tgt, _ = O.gen_demos_i8(B, S, K, thr, (-1, 0, 1), 1, seed=S + K)
    st = tgt.copy()
    ac = tok.copy()
    # 0..5: plain replays: done exactly at K-1 (or earlier if terms cancel)
    # 6..8: the last two actions are a term and its negation: the state is zero at K-3 already
    for b in (6, 7, 8):
        ac[b, K - 2] = tok[b, 0]
        ac[b, K - 1] = tok[b, 0]
        ac[b, K - 1, :S] = 2 - tok[b, 0, :S]
        st[b] = O.gen_from_factors_i8(ac[b:b + 1, :K - 2], 1)[0][0]
    # 9..11: never done (random start state)
    st[9:12] = rng.integers(-3, 4, size=(3, S, S, S))
    # 12: starts at zero, first action is null -> done at step 0, then leaves zero
    st[12] = 0
    ac[12, 0] = 1
    # 13: a dense action pushes entries beyond int8 in the middle, its negation brings them back
    ac[13, 3] = 2
    ac[13, 3, 2 * S:] = 2
    st[13] = np.clip(st[13].astype(int) - 126, -128, 127).astype(np.int8)
    ac[13, 4] = ac[13, 3]
    ac[13, 4, :S] = 0
    # 14: factors beyond the byte-product range (|u| = 12)
    ac[14, 1, 0] = 13
    # 15: full int8 w
    ac[15, 2, 2 * S:] = rng.integers(-127, 128, size=S)
    # 16..17: dense {-2..2} factors: the scalar bound fails; the elementwise bound certifies short lists, the lattice
    # kernels the rest
    ac[16:18] = rng.integers(-1, 4, size=(2, K, 3 * S))
    # 18: passes through zero at step 5 and leaves again
    st[18] = O.gen_from_factors_i8(ac[18:19, :6], 1)[0][0]
    want, want_ds, want_ovf = O.step_many_i8(st, ac)
    assert (want_ds[:6] >= 0).all() and (want_ds[6:9] == K - 3).all() and (want_ds[9:12] == -1).all()
    assert want_ds[12] == 0 and want_ovf[13] == 1 and want_ds[18] == 5
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    out, ds = ops.step_many(padded(st), dev(ac), overflow=ovf)
    assert np.array_equal(host(out), want) and np.array_equal(host(ds), want_ds) and np.array_equal(host(ovf), want_ovf)
    # in place
    t = padded(st)
    ovf.zero_()
    ops.step_many(t, dev(ac), out=t, overflow=ovf)
    assert np.array_equal(host(t), want) and np.array_equal(host(ovf), want_ovf)


@pytest.mark.parametrize("S,K,values,probs", [
    (25, 64, (-2, -1, 0, 1, 2), (0.05, 0.1, 0.7, 0.1, 0.05)),   # the paper's vocabulary at BASELINE config 5's length
    (16, 64, (-2, -1, 0, 1, 2), (0.05, 0.1, 0.7, 0.1, 0.05)),
    (25, 200, (-1, 0, 1), (0.15, 0.7, 0.15)),                   # K > 127: beyond the scalar bound for any vocabulary
    (16, 256, (-1, 0, 1), (0.15, 0.7, 0.15)),
    (9, 130, (-1, 0, 1), (0.15, 0.7, 0.15)),
])
def test_step_many_elementwise_bound_keeps_games_on_the_matrix_cores(S, K, values, probs):
    """the scalar overflow bound (max|final| + sum of max|u| max|v| max|w|) hands every {-2..2} game and
    every list beyond 127 actions to the lattice kernels; the elementwise bound |X0| + sum |u||v||w| certifies them on the
    matrix cores.  Demo replays: exact results, and NOT ONE game handed over."""
    B = 40
    thr = O.categorical_thresholds(probs)
    tok, tgt, ovf0 = O.gen_demos_i8(B, S, K, thr, values, 1, seed=3 * S + K)
    keep = ovf0 == 0
    tok, tgt = tok[keep], tgt[keep]
    assert len(tok) >= B // 2
    want, want_ds, want_ovf = O.step_many_i8(tgt, tok)
    assert not want.any() and not want_ovf.any()
    before_h, before_f = ops.debug_handovers(DEV), ops.debug_fallbacks(DEV)
    ovf = torch.zeros(len(tok), dtype=torch.uint8, device=DEV)
    out, ds = ops.step_many(padded(tgt), dev(tok), overflow=ovf)
    assert np.array_equal(host(out), want) and np.array_equal(host(ds), want_ds) and not host(ovf).any()
    early = int((want_ds != K - 1).sum())           # a replay whose last terms cancel is done early: that one is redone
    assert ops.debug_handovers(DEV) - before_h == early and early <= 2
    assert ops.debug_fallbacks(DEV) == before_f
    # and a game that does overflow in the middle is still caught: dense +-2 action and its negation
    tok2, tgt2 = tok[:4].copy(), tgt[:4].copy()
    tok2[:, 1] = -1                                   # u = v = w = -2 everywhere: every entry + 8
    tok2[:, 2] = tok2[:, 1]
    tok2[:, 2, :S] = 3                                # ... and back
    tgt2[:] = 125
    w2, wds2, wovf2 = O.step_many_i8(tgt2, tok2)
    assert wovf2.all()
    ovf2 = torch.zeros(4, dtype=torch.uint8, device=DEV)
    o2, ds2 = ops.step_many(padded(tgt2), dev(tok2), overflow=ovf2)
    assert np.array_equal(host(o2), w2) and np.array_equal(host(ds2), wds2) and np.array_equal(host(ovf2), wovf2)


@pytest.mark.parametrize("S", [9, 16, 25])
@pytest.mark.parametrize("R", [1, 31, 32, 33, 64, 65, 100, 256, 257])
def test_gen_from_factors_matrix_core_path(S, R):
    """tg_mfma.h: R around the 32-action k-steps (compile-time 1 and 2 steps, run-time loop, the
    R > 256 hand-over to the vector kernels), factors at and beyond the |u|,|v| <= 11 bound of the
    int8 byte products (games beyond it take the exact byte-wise form), full-range w, overflow."""
    rng = np.random.default_rng(1000 * S + R)
    B = 13
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, R, 3 * S)).astype(np.int8)
    ac[1] = rng.integers(-10, 13, size=(R, 3 * S))          # |u|,|v|,|w| <= 11 exactly at the bound
    ac[2] = ac[1]
    ac[2, R // 2, 3] = 13                                    # one u = 12: this game leaves the fast path
    ac[3, :, 2 * S:] = rng.integers(-127, 128, size=(R, S))  # w = token - 1 down to -128: the whole int8 range
    ac[4, :, :2 * S] = rng.integers(-10, 13, size=(R, 2 * S))
    ac[5] = 1                                                # all-zero factors
    ac[6, :, :] = rng.choice([0, 2], size=(R, 3 * S))        # +-1 only: sums up to R
    want, wovf = O.gen_from_factors_i8(ac, 1)
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    before = ops.debug_fallbacks()
    out = ops.gen_from_factors(dev(ac), S, overflow=ovf)
    assert np.array_equal(host(out), want) and np.array_equal(host(ovf), wovf)
    import os
    if R <= 256 and not os.environ.get("TG_NO_MFMA"):        # (the vector kernels have other factor limits)
        assert ops.debug_fallbacks() - before == 1           # game 2 only
    # padded game stride and a second shift
    want2, wovf2 = O.gen_from_factors_i8(ac, 2)
    buf = ops.alloc_states(B, S, DEV)
    ovf.zero_()
    ops.gen_from_factors(dev(ac), S, out=buf, overflow=ovf, shift=2)
    assert np.array_equal(host(buf), want2) and np.array_equal(host(ovf), wovf2)


@pytest.mark.parametrize("S,R", [(9, 5), (16, 33), (25, 64), (25, 70)])
def test_basis_tokens_matrix_core_path_large_entries(S, R):
    """Change of basis on the tokens with basis entries over the whole int8 range (not unimodular: the
    kernel does not care) and shift 2: wrapped tokens + overflow flag must equal the oracle's."""
    rng = np.random.default_rng(77 * S + R)
    B = 9
    P = rng.integers(-3, 4, size=(B, 3, S, S)).astype(np.int8)
    P[0] = rng.integers(-128, 128, size=(3, S, S))
    P[1] = np.eye(S, dtype=np.int8)
    thr = O.categorical_thresholds((0.15, 0.7, 0.15))
    tok_o, tgt_o, ovf_o = O.gen_demos_i8(B, S, R, thr, (-1, 0, 1), 2, seed=5, basis=P)
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    tok, tgt = ops.gen_demos(B, S, R, DEV, seed=5, shift=2, basis=dev(P), overflow=ovf)
    assert np.array_equal(host(tok), tok_o) and np.array_equal(host(tgt), tgt_o)
    assert np.array_equal(host(ovf), ovf_o) and ovf_o[0] == 1 and ovf_o[1] == 0


@pytest.mark.parametrize("S,B,K", [(4, 130, 7), (4, 5, 40), (9, 33, 12), (16, 9, 70), (25, 3, 130), (5, 6, 9), (2, 3, 3),
                                   (25, 6, 3), (25, 5, 5), (25, 9, 2), (16, 11, 20), (16, 7, 19), (9, 13, 30), (9, 8, 29)])
def test_step_many_matches_oracle(S, B, K):
    rng = np.random.default_rng(S * 31 + K)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, K, 3 * S)).astype(np.int8)
    st = rng.integers(-1, 2, size=(B, S, S, S)).astype(np.int8)
    for b in range(0, B, 2):  # these games hit zero exactly after step b % K
        kk = b % K
        st[b] = O.gen_from_factors_i8(ac[b:b + 1, :kk + 1])[0][0]
    want, want_ds, want_ovf = O.step_many_i8(st, ac)
    for t in (padded(st), dev(st)):
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, ds = ops.step_many(t, dev(ac), overflow=ovf)
        assert np.array_equal(host(out), want) and np.array_equal(host(ds), want_ds), (S, B, K)
        assert np.array_equal(host(ovf), want_ovf)
    assert (want_ds >= 0).sum() >= B // 2
    # K single steps == one step_many
    t = padded(st)
    for k in range(K):
        ops.step(t, dev(ac[:, k]), out=t)
    assert np.array_equal(host(t), want)


@pytest.mark.parametrize("S,B,K", [(4, 40, 6), (9, 13, 5), (9, 13, 2), (16, 10, 9), (25, 5, 7), (25, 5, 2), (6, 5, 4)])
def test_step_many_overflow_and_wide_factors(S, B, K):
    """step_many where some games overflow int8 mid-way (the lattice form must hand them to the exact
    path), some use factors too large for the 16-bit path, and some are ordinary."""
    rng = np.random.default_rng(S * 7 + K)
    st = rng.integers(-3, 4, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.2, 0.6, 0.2], size=(B, K, 3 * S)).astype(np.int8)
    st[0::4] = rng.choice([-128, -127, 126, 127], size=st[0::4].shape).astype(np.int8)   # overflow within a few steps
    ac[0::4] = rng.choice([0, 2], size=ac[0::4].shape)                                  # dense +-1 factors
    ac[1::4, K // 2] = rng.integers(-6, 9, size=ac[1::4, K // 2].shape)                   # |factor| up to 7
    st[2::4] = 0
    st[2::4, 0, 0, 0] = 127                                                              # saturating edge: 127 -/+ 1
    want, want_ds, want_ovf = O.step_many_i8(st, ac)
    assert want_ovf[0::4].all() and not want_ovf[3::4].any()
    for t in (padded(st), dev(st)):
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, ds = ops.step_many(t, dev(ac), overflow=ovf)
        assert np.array_equal(host(out), want), (S, K)
        assert np.array_equal(host(ds), want_ds) and np.array_equal(host(ovf), want_ovf)
    t = padded(st)                                                                        # in place
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    ops.step_many(t, dev(ac), out=t, overflow=ovf)
    assert np.array_equal(host(t), want) and np.array_equal(host(ovf), want_ovf)


@pytest.mark.parametrize("S,B,k", [(4, 67, 8), (9, 5, 5), (16, 6, 3), (25, 2, 4), (3, 4, 2), (4, 1, 130)])
def test_expand_matches_oracle(S, B, k):
    rng = np.random.default_rng(S + 100 * k)
    st, ac = rand_case(rng, B, S, k=k)
    ac[1 % B, k - 1, :S] = 1  # null action (u = 0)
    want, want_done, want_chg, _ = O.expand_i8(st, ac)
    for t in (padded(st), dev(st)):
        kids, done, chg = ops.expand(t, dev(ac))
        assert np.array_equal(host(kids), want) and np.array_equal(host(done), want_done)
        assert np.array_equal(host(chg), want_chg)
    assert want_done.any() and not want_chg.all()


def test_done_and_nnz():
    rng = np.random.default_rng(5)
    for S, B in [(4, 300), (9, 50), (16, 20), (25, 7), (7, 11)]:
        st = (rng.random((B, S, S, S)) < 0.02).astype(np.int8) * rng.integers(-3, 4, size=(B, S, S, S)).astype(np.int8)
        st[::3] = 0
        for t in (padded(st), dev(st)):
            d, nnz = ops.done(t, want_nnz=True)
            assert np.array_equal(host(d), O.done_per_game(st).astype(np.uint8))
            assert np.array_equal(host(nnz), O.nnz_per_game(st))


# ------------------------------------------------------------------ generator
@pytest.mark.parametrize("S,B,R", [(4, 300, 7), (9, 40, 12), (16, 12, 20), (25, 5, 30), (5, 9, 4)])
def test_generator_bit_exact(S, B, R):
    probs, values = (0.15, 0.7, 0.15), (-1, 0, 1)
    thr = O.categorical_thresholds(probs)
    want_tok, want_tgt, want_ovf = O.gen_demos_i8(B, S, R, thr, values, 1, seed=1234, game_id_offset=1000)
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    tok, tgt = ops.gen_demos(B, S, R, DEV, values=values, probs=probs, seed=1234, game_id_offset=1000, overflow=ovf)
    assert np.array_equal(host(tok), want_tok) and np.array_equal(host(tgt), want_tgt)
    assert np.array_equal(host(ovf), want_ovf)
    # sharding invariance: two shards == one batch (section 8e)
    h = B // 3
    a_tok, a_tgt = ops.gen_demos(h, S, R, DEV, seed=1234, game_id_offset=1000)
    b_tok, b_tgt = ops.gen_demos(B - h, S, R, DEV, seed=1234, game_id_offset=1000 + h)
    assert np.array_equal(np.concatenate([host(a_tok), host(b_tok)]), want_tok)
    assert np.array_equal(np.concatenate([host(a_tgt), host(b_tgt)]), want_tgt)
    # other distributions / shift, more categories
    thr5 = O.categorical_thresholds((1, 2, 10, 2, 1))
    w_tok, w_tgt, _ = O.gen_demos_i8(B, S, R, thr5, (-2, -1, 0, 1, 2), 2, seed=7)
    g_tok, g_tgt = ops.gen_demos(B, S, R, DEV, values=(-2, -1, 0, 1, 2), probs=(1, 2, 10, 2, 1), shift=2, seed=7)
    assert np.array_equal(host(g_tok), w_tok) and np.array_equal(host(g_tgt), w_tgt)


@pytest.mark.parametrize("S,B,R,values,probs", [
    (9, 21, 200, (-2, -1, 0, 1, 2), (3, 2, 1, 2, 3)),      # dense +-2 factors, long lists: targets overflow (range-checked tiles)
    (16, 9, 130, (-1, 0, 1), (0.15, 0.7, 0.15)),           # ternary but R > 127: the bound R * f^3 <= 127 fails
    (25, 6, 127, (-1, 0, 1), (0.3, 0.4, 0.3)),             # ternary, R = 127: the unchecked tiles, at their limit
    (25, 5, 256, (-1, 0, 1), (0.15, 0.7, 0.15)),           # the largest R of the fused kernel
    (16, 7, 33, (0, 3), (0.5, 0.5)),                        # two categories (one threshold), tokens 1 and 4
    (9, 30, 8, (-11, 0, 11), (0.2, 0.6, 0.2)),              # factors at the byte-product limit of the matrix-core path
    (9, 30, 8, (-12, 0, 12), (0.2, 0.6, 0.2)),              # one beyond it: token kernel + tg_gen_from_factors_i8
])
def test_generator_overflow_flags_and_vocabularies(S, B, R, values, probs):
    """tg_gen_demos_i8 against the oracle where the fused kernel switches variants: range-checked and unchecked tiles,
    run-time k-loop, general (non-ternary) draw evaluation, and the fall-back to the two-kernel path."""
    thr = O.categorical_thresholds(probs)
    want_tok, want_tgt, want_ovf = O.gen_demos_i8(B, S, R, thr, values, 1, seed=77, game_id_offset=5)
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    tok, tgt = ops.gen_demos(B, S, R, DEV, values=values, probs=probs, seed=77, game_id_offset=5, overflow=ovf)
    assert np.array_equal(host(tok), want_tok) and np.array_equal(host(tgt), want_tgt)
    assert np.array_equal(host(ovf), want_ovf)


def test_generator_distribution_vs_reference(golden):
    from scipy import stats
    g = golden("sampler_stats")
    for S, probs, key in [(4, (0.15, 0.7, 0.15), "S4_p70"), (9, (0.15, 0.7, 0.15), "S9_p70"), (4, (0.1, 0.8, 0.1), "S4_p80")]:
        tok, _ = ops.gen_demos(4096, S, 3, DEV, probs=probs, seed=2026)
        f = host(tok).astype(np.int64) - 1
        assert (f.reshape(-1, S) != 0).any(axis=1).all()
        ours = np.array([(f == v).sum() for v in (-1, 0, 1)])
        _, p, _, _ = stats.chi2_contingency(np.stack([ours, g[key + "_value_counts"]]))
        assert p > 1e-3, (key, ours, p)


def test_synthetic_demos_generate_entry():
    """SyntheticDemos.generate == the constructor; random_basis emits every demo in its own
    GL(S,Z) basis and the demo still replays to zero."""
    a = SyntheticDemos.generate(n_demos=5, dim_3d=9, max_actions=6, device=DEV, seed=3)
    b = SyntheticDemos(max_actions=6, n_demos=5, dim_t=1, dim_3d=9, device=DEV, seed=3)
    assert torch.equal(a.action_seq, b.action_seq) and torch.equal(a.target_tensor, b.target_tensor)
    c = SyntheticDemos.generate(n_demos=5, dim_3d=9, max_actions=6, device=DEV, seed=3, random_basis=True)
    assert not torch.equal(c.action_seq, a.action_seq)
    final, ds = ops.step_many(c.target_tensor, c.action_seq)
    ok = host(c.overflow) == 0
    assert ok.any() and not host(final)[ok].any() and (host(ds)[ok] >= 0).all()


def test_synthetic_demos_class_matches_reference_framing(golden):
    """SyntheticDemos.batch == the reference's __getitem__ arithmetic on the same tokens."""
    demos = SyntheticDemos(max_actions=6, n_demos=9, dim_t=3, dim_3d=4, device=DEV, seed=11)
    tok, tgt = host(demos.action_seq), host(demos.target_tensor)
    for ia in range(6):
        state, scalar, action, reward = demos.batch(ia)
        for d in (0, 4, 8):
            frames, sc, act, rw = O.demo_getitem(list(tok[d]), tgt[d], ia, 3)
            assert np.array_equal(host(state)[d], frames)
            assert float(scalar[d]) == sc and float(reward[d]) == rw and np.array_equal(host(action)[d], act)
    seq, target = F.create_synthetic_demo((-1, 0, 1), (0.1, 0.8, 0.1), 5, 4, 1, seed=3, device=DEV)
    w_tok, w_tgt, _ = O.gen_demos_i8(1, 4, 5, O.categorical_thresholds((0.1, 0.8, 0.1)), (-1, 0, 1), 1, seed=3)
    assert np.array_equal(np.stack([host(a) for a in seq]), w_tok[0]) and np.array_equal(host(target), w_tgt[0])


# ------------------------------------------------------------------ change of basis (A12, parity unpinned: invariants)
@pytest.mark.parametrize("S", [4, 9, 16, 25, 6, 32])
def test_basis(S):
    B = 7 if S < 32 else 2   # S=32 needs 135 KB of LDS per workgroup (dynamic LDS above 64 KB)
    bp = (0.2, 0.6, 0.2) if S <= 9 else (0.03, 0.94, 0.03)
    thr = O.categorical_thresholds(bp)
    P_o, L_o, U_o = O.sample_basis(B, S, thr, (-1, 0, 1), seed=21, game_id_offset=5)
    P, L, U = ops.sample_basis(B, S, DEV, probs=bp, seed=21, game_id_offset=5, want_factors=True)
    assert np.array_equal(host(P), P_o) and np.array_equal(host(L), L_o) and np.array_equal(host(U), U_o)
    fthr = O.categorical_thresholds((0.15, 0.7, 0.15))
    R = 4
    tok_o, tgt_o, ovf_o = O.gen_demos_i8(B, S, R, fthr, (-1, 0, 1), 1, seed=9, basis=P_o)
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    tok, tgt = ops.gen_demos(B, S, R, DEV, seed=9, basis=P, overflow=ovf)
    assert np.array_equal(host(tok), tok_o) and np.array_equal(host(tgt), tgt_o) and np.array_equal(host(ovf), ovf_o)
    # tensor-level change of basis == oracle, == factor-level transform, inverse restores
    plain_tok, plain_tgt = ops.gen_demos(B, S, R, DEV, seed=9)
    ovf2 = torch.zeros(B, dtype=torch.uint8, device=DEV)
    moved = ops.change_basis(plain_tgt, P.to(torch.int32), overflow=ovf2)
    moved_o, movf_o = O.change_basis_i8(host(plain_tgt), P_o)
    assert np.array_equal(host(moved), moved_o) and np.array_equal(host(ovf2), movf_o)
    ok = (movf_o == 0) & (ovf_o == 0)
    assert ok.any() and np.array_equal(host(moved)[ok], tgt_o[ok])
    Pinv = O.unimodular_inverse(L_o, U_o)
    if np.abs(Pinv).max() < 2 ** 31:
        back = ops.change_basis(moved, dev(Pinv.astype(np.int32)))
        assert np.array_equal(host(back)[ok], host(plain_tgt)[ok])
    eye = torch.eye(S, dtype=torch.int32, device=DEV).expand(B, 3, S, S).contiguous()
    assert np.array_equal(host(ops.change_basis(plain_tgt, eye)), host(plain_tgt))


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
@pytest.mark.parametrize("S,B,R", [(4, 65536, 7), (16, 8192, 20), (25, 4096, 64), (9, 20000, 12),
                                   (4, 131072, 7), (4, 1 << 20, 7),    # cfg4: per-GPU share at 8 GPUs, and whole
                                   (16, 32768, 12), (9, 262144, 6),    # S=16 with whole-line stores (128 MiB); S=9 streaming
                                   (16, 98304, 5),                     # S=16 beyond the cache (384 MiB of states): non-temporal state loads
                                   (25, 8192, 6), (25, 26000, 4),      # S=25 with whole-line stores (122 MiB) / + non-temporal loads (388 MiB)
                                   (4, 1 << 21, 4)])                   # S=4 beyond 96 MiB: non-temporal state loads
def test_full_size_generate_replay_terminate(S, B, R):
    """cfg2/cfg3/cfg4/cfg5-per-GPU: generator -> replay the demo's own actions -> every game reaches
    zero exactly at the last step (generator, step, step_many and done agree)."""
    demos = SyntheticDemos(R, B, 1, S, DEV, seed=S)
    assert not bool(demos.overflow.any())
    rev = demos.action_seq.flip(1).contiguous()
    # track_nnz=False: tg_step_i8 in the variant this footprint selects (whole lines, non-temporal loads, ...); then the env
    # as a user gets it (from 12 000 games at S=16 / 2 048 at S=25 it carries nnz and takes the tracked step)
    for track in (False, None):
        env = TensorGameEnv(B, S, DEV, track_nnz=track)
        env.reset(demos.target_tensor)
        for k in range(R):
            state, done = env.step(rev[:, k])   # (a game may reach zero early only if the remaining terms cancel)
        assert bool(done.all()) and not bool(state.any()) and not env.any_overflow()
        assert int(env.nnz().sum()) == 0
    env.reset(demos.target_tensor)
    state, done_step = env.step_many(demos.action_seq)
    assert not bool(state.any()) and bool((done_step >= 0).all()) and int(done_step.max()) == R - 1
    # linearity: target(a ++ b) == target(a) + target(b) (checksum of sums, no overflow at these R)
    half = R // 2
    ta = ops.gen_from_factors(demos.action_seq[:, :half].contiguous(), S)
    tb = ops.gen_from_factors(demos.action_seq[:, half:].contiguous(), S)
    assert torch.equal((ta.to(torch.int16) + tb.to(torch.int16)).to(torch.int8), demos.target_tensor)
    # order independence of _take_actions (datasets.py:144-153 is a pure sum)
    perm = torch.randperm(R, device=DEV)
    env.reset(demos.target_tensor)
    state, _ = env.step_many(demos.action_seq[:, perm].contiguous())
    assert not bool(state.any())


@pytest.mark.parametrize("S,B,R", [(25, 88000, 3),       # 1.28 GiB of S=25 states: three workgroups per CU (unused dynamic LDS), plain loads
                                   (16, 336000, 3),      # 1.28 GiB of S=16 states: whole lines, five workgroups per CU
                                   (4, 6300000, 3)])     # 385 MiB of S=4 states: non-temporal loads + the token awaited first
def test_launch_shapes_beyond_the_caches_replay_to_zero(S, B, R):
    """the launch shapes tg_step_i8 takes only by FOOTPRINT (occupancy held down by unused dynamic LDS from 1.25 GiB
    on, the S=4 token wait from 384 MiB) had their arithmetic variants forced against the oracle at small batches, but the
    launches themselves ran at full size only under bench.py's self-check.  Here: generate, replay the demonstration's own
    actions in reverse with the plain step (and with the tracked one the env takes by itself) -- every game must be all
    zero exactly at the last step, twice (both sweep directions), and step_many must agree."""
    demos = SyntheticDemos(R, B, 1, S, DEV, seed=S + 1)
    rev = demos.action_seq.flip(1).contiguous()
    for track in (False, None):
        env = TensorGameEnv(B, S, DEV, track_nnz=track)
        for _ in range(2):
            env.reset(demos.target_tensor)
            for k in range(R):
                state, done = env.step(rev[:, k])
                if k < R - 1:
                    assert not bool(done.all())
            assert bool(done.all()) and not bool(state.any()) and not env.any_overflow()
        del env
        torch.cuda.empty_cache()
    final, done_step = ops.step_many(demos.target_tensor, demos.action_seq)
    assert not bool(final.any()) and int(done_step.max()) == R - 1 and int(done_step.min()) >= 0


def test_sharded_env_equals_single_env():
    S, B, R = 4, 1000, 7
    demos = SyntheticDemos(R, B, 1, S, DEV, seed=4)
    full = TensorGameEnv(B, S, DEV)
    full.reset(demos.target_tensor)
    full.step(demos.action_seq[:, 0])
    parts = []
    for rank in range(3):
        lo, hi = mat_mul_amd.shard_range(B, rank, 3)
        sd = SyntheticDemos.sharded(R, B, 1, S, rank, 3, DEV, seed=4)
        assert sd.game_id_offset == lo and sd.n_demos == hi - lo
        assert torch.equal(sd.action_seq, demos.action_seq[lo:hi])
        e = TensorGameEnv.sharded(B, S, rank, 3, DEV)
        assert e.game_id_offset == lo and e.B == hi - lo
        e.reset(sd.target_tensor)
        e.step(sd.action_seq[:, 0])
        parts.append(e.state)
    assert torch.equal(torch.cat(parts), full.state)


@pytest.mark.parametrize("S,T", [(4, 1), (4, 3), (16, 2)])
def test_env_graph_stepper_equals_eager_steps(S, T):
    """TensorGameEnv.graph_stepper: replayed hipGraphs over a static token buffer == eager step() calls,
    including the history ring (one graph per slot)."""
    B, K = 37, 7
    demos = SyntheticDemos(max_actions=K, n_demos=B, dim_t=T, dim_3d=S, device=DEV, seed=5)
    eager = TensorGameEnv(B, S, DEV, dim_t=T)
    graph = TensorGameEnv(B, S, DEV, dim_t=T)
    eager.reset(demos.target_tensor)
    graph.reset(demos.target_tensor)
    buf = torch.empty((B, 3 * S), dtype=torch.int8, device=DEV)
    step = graph.graph_stepper(buf)
    for k in reversed(range(K)):
        buf.copy_(demos.action_seq[:, k])
        s1, d1 = step()
        s0, d0 = eager.step(demos.action_seq[:, k].contiguous())
        assert torch.equal(s1, s0) and torch.equal(d1, d0) and graph.head == eager.head
        assert torch.equal(graph.model_input()[0], eager.model_input()[0])
    assert bool(d1.all())


def test_graph_capture_of_steps():
    """Launches go to the caller's stream, so K steps can be captured in one hipGraph."""
    S, B, R = 4, 4096, 7
    demos = SyntheticDemos(R, B, 1, S, DEV, seed=8)
    env = TensorGameEnv(B, S, DEV)
    env.reset(demos.target_tensor)
    acts = [demos.action_seq[:, k].contiguous() for k in range(R)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step(acts[0])          # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    env.reset(demos.target_tensor)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for k in range(R):
            env.step(acts[k])
    env.reset(demos.target_tensor)
    graph.replay()
    torch.cuda.synchronize()
    assert bool(env.done.all()) and not bool(env.state.any())


# ------------------------------------------------------------------ next rows: N1 history / model input, N2 hash, N3 rank
def test_history_ring_and_model_input_golden(golden):
    g = golden("next_rows")
    for tag, S, T in (("hist_S4_T3", 4, 3), ("hist_S9_T2", 9, 2)):
        states, acts = g[tag + "_states"], g[tag + "_actions"]
        B = states.shape[1]
        env = TensorGameEnv(B, S, DEV, dim_t=T)
        env.reset(dev(states[0]))
        x, sc = env.model_input()
        assert np.array_equal(host(x), states[0].astype(np.float32))
        for k in range(len(acts)):
            env.step(dev(acts[k]))
            for dt in (torch.float32, torch.float16, torch.bfloat16):
                x, sc = env.model_input(dt)
                assert x.dtype == dt and np.array_equal(host(x.float()), states[k + 1].astype(np.float32)), (tag, k)
            assert host(sc).tolist() == [[float(k + 1)]] * B
    # fresh games: zero history (build_matmul_tensor(dim_t, ...), utils.py:157), odd sizes / unaligned frames
    env = TensorGameEnv(5, 9, DEV, dim_t=4)
    env.reset()
    x, _ = env.model_input()
    want = np.broadcast_to(O.build_matmul_tensor(4, 3, 3, 3), (5, 4, 9, 9, 9)).astype(np.float32)
    assert np.array_equal(host(x), want)
    ring = torch.randint(-5, 6, (3, 2, 5, 5, 5), dtype=torch.int8, device=DEV)  # packed 125-byte frames
    x, _ = ops.emit_frames(ring, 1, 0.0)
    assert np.array_equal(host(x), host(ring)[:, ::-1].astype(np.float32))
    # the whole int8 range is exact in every output type (bfloat16 has 8 significant bits: |x| <= 256)
    ring = torch.arange(-128, 128, dtype=torch.int16, device=DEV).to(torch.int8).repeat(4)[:1000].reshape(1, 1, 10, 10, 10)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x, _ = ops.emit_frames(ring.contiguous(), 0, 0.0, dt)
        assert x.dtype == dt and torch.equal(x.float(), ring.float())


def test_state_hash_matches_oracle(golden):
    rng = np.random.default_rng(3)
    for S, B in [(4, 448), (9, 33), (16, 9), (25, 5), (5, 17), (1, 3)]:
        st = rng.integers(-3, 4, size=(B, S, S, S)).astype(np.int8)
        if S == 4:
            st = golden("strassen")["ds_states"]
        want = O.state_hash(st)
        for t in (padded(st), dev(st)):
            got = host(ops.state_hash(t)).view(np.uint64)
            assert np.array_equal(got, want), S
    s = golden("strassen")["ds_states"]
    keys = host(ops.state_hash(padded(s)))
    strings = ["_".join(map(str, x.reshape(-1))) for x in s]
    assert len(set(keys.tolist())) == len(set(strings))


def test_slice_rank_golden_and_oracle(golden):
    g = golden("next_rows")
    for S in (4, 9, 16, 25):
        st = g[f"rank_S{S}_state"]
        got = host(ops.slice_rank(padded(st)))
        assert np.array_equal(got, g[f"rank_S{S}_rank"]), S          # == reference get_rank per game
        assert np.array_equal(got, O.slice_rank_exact(st))
    rng = np.random.default_rng(9)
    for S in (3, 6, 32):
        st = rng.integers(-2, 3, size=(4, S, S, S)).astype(np.int8)
        st[0] = 0
        st[1, :, 1:] = st[1, :, :1]                                  # every slice has rank <= 1
        assert np.array_equal(host(ops.slice_rank(dev(st))), O.slice_rank_exact(st))
    # rank-deficient slices with entries over the whole int8 range (the kernel eliminates modulo two 26-bit primes in
    # double precision: huge minors, exact zeros required) against the exact fraction-free oracle
    for S, B in [(25, 6), (16, 10), (9, 24), (5, 40)]:
        st = np.zeros((B, S, S, S), np.int64)
        for b in range(B):
            for i in range(S):
                for _ in range(int(rng.integers(0, 5))):                 # slice i = sum of up to 4 rank-1 matrices
                    st[b, i] += np.outer(rng.integers(-6, 7, S), rng.integers(-6, 7, S))
        st = np.clip(st, -128, 127).astype(np.int8)
        st[0] = rng.integers(-128, 128, size=(S, S, S))                  # full rank, large entries
        assert np.array_equal(host(ops.slice_rank(padded(st))), O.slice_rank_exact(st)), S
    env = TensorGameEnv(3, 4, DEV)
    env.reset()
    assert host(env.rank_reward()).tolist() == [-8, -8, -8]           # <2,2,2>: four slices of rank 2


@pytest.mark.parametrize("S,B,K", [(9, 12, 4), (16, 8, 5), (25, 4, 3), (9, 6, 2)])
def test_step_many_moderate_factors_stay_on_fast_path(S, B, K):
    """Factors up to +-11 (single updates of up to 1331 on sparse vectors) with states that stay in int8:
    the saturating lattice form must be exact, and must not need the fallback."""
    rng = np.random.default_rng(S + K)
    ac = np.ones((B, K, 3 * S), np.int8)                                   # factor 0 everywhere (shift 1)
    for b in range(B):
        for k in range(K):
            for part, lo, hi in ((0, -11, 12), (1, -11, 12), (2, -1, 2)):      # one non-zero u_i, v_j; w in {-1,0,1}
                if part < 2:
                    ac[b, k, part * S + rng.integers(S)] = 1 + rng.integers(lo, hi)
                else:
                    ac[b, k, 2 * S:] = 1 + rng.integers(lo, hi, size=S)
    st = rng.integers(-3, 4, size=(B, S, S, S)).astype(np.int8)
    want, want_ds, want_ovf = O.step_many_i8(st, ac)
    # keep only games whose path stays in range at every step (the others legitimately fall back)
    keep = want_ovf == 0
    assert keep.sum() >= 1
    st, ac, want, want_ds = st[keep], ac[keep], want[keep], want_ds[keep]
    before = ops.debug_fallbacks(DEV)
    ovf = torch.zeros(len(st), dtype=torch.uint8, device=DEV)
    out, ds = ops.step_many(padded(st), dev(ac), overflow=ovf)
    assert np.array_equal(host(out), want) and np.array_equal(host(ds), want_ds) and not host(ovf).any()
    assert np.abs(ac.astype(int) - 1).max() > 5
    assert ops.debug_fallbacks(DEV) == before


def test_default_vocabulary_never_falls_back():
    """The packed/rows kernels silently fall back to a 10-50x slower exact form for out-of-range
    factors or int8 overflow; ordinary inputs ({-1,0,1} and {-2..2} factors, no overflow) must not."""
    before = ops.debug_fallbacks(DEV)
    for S, B, R in [(9, 50, 7), (16, 30, 9), (25, 9, 12), (9, 20, 2), (25, 6, 2)]:
        for values, probs in [((-1, 0, 1), (0.15, 0.7, 0.15)), ((-2, -1, 0, 1, 2), (0.05, 0.1, 0.7, 0.1, 0.05))]:
            tok, tgt = ops.gen_demos(B, S, R, DEV, values=values, probs=probs, seed=5)
            st = ops.alloc_states(B, S, DEV)
            st.copy_(tgt)
            ops.step(st, tok[:, 0].contiguous())
            ops.expand(st, tok[:, :2].contiguous())
            out, ds = ops.step_many(st, tok)
            assert not bool(out.any())
    assert ops.debug_fallbacks(DEV) == before
    st = torch.full((4, 16, 16, 16), 127, dtype=torch.int8, device=DEV)          # forces the overflow path
    ac = torch.full((4, 3, 48), 0, dtype=torch.int8, device=DEV)                 # factors -1: 127 - (-1) overflows
    ops.step_many(st, ac)
    assert ops.debug_fallbacks(DEV) > before


# ------------------------------------------------------------------ edge cases: empty batch, maximum action counts
def test_empty_batch_is_a_noop():
    for S in (4, 16, 25, 7):
        st = ops.alloc_states(0, S, DEV)
        ac = torch.empty((0, 3 * S), dtype=torch.int8, device=DEV)
        out, done = ops.step(st, ac)
        assert out.shape == (0, S, S, S) and done.shape == (0,)
        out, ds = ops.step_many(st, torch.empty((0, 5, 3 * S), dtype=torch.int8, device=DEV))
        assert ds.shape == (0,)
        kids, d, c = ops.expand(st, torch.empty((0, 3, 3 * S), dtype=torch.int8, device=DEV))
        assert kids.shape == (0, 3, S, S, S)
        assert ops.done(st).shape == (0,) and ops.state_hash(st).shape == (0,) and ops.slice_rank(st).shape == (0,)
        tok, tgt = ops.gen_demos(0, S, 4, DEV)
        assert tok.shape == (0, 4, 3 * S) and tgt.shape == (0, S, S, S)
    env = TensorGameEnv(0, 4, DEV)
    env.reset()
    assert env.model_input()[0].shape == (0, 1, 4, 4, 4)


@pytest.mark.parametrize("S,B", [(4, 3), (9, 2), (16, 2), (25, 1), (6, 1)])
def test_maximum_action_counts(S, B):
    """K = k = R = TG_MAX_ACTIONS (4096) and counts that straddle the LDS action tiles."""
    from mat_mul_amd._lib import TG_MAX_ACTIONS
    rng = np.random.default_rng(S)
    for K in (TG_MAX_ACTIONS, 65, 129):
        ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, K, 3 * S)).astype(np.int8)
        tgt_o, ovf_o = O.gen_from_factors_i8(ac)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        tgt = ops.gen_from_factors(dev(ac), S, overflow=ovf)
        assert np.array_equal(host(tgt), tgt_o) and np.array_equal(host(ovf), ovf_o)
        st = rng.integers(-1, 2, size=(B, S, S, S)).astype(np.int8)
        want, want_ds, want_ovf = O.step_many_i8(st, ac)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        out, ds = ops.step_many(padded(st), dev(ac), overflow=ovf)
        assert np.array_equal(host(out), want) and np.array_equal(host(ds), want_ds) and np.array_equal(host(ovf), want_ovf)
        if K <= 129 or S <= 9:
            kids_o, done_o, chg_o, _ = O.expand_i8(st, ac)
            kids, done, chg = ops.expand(padded(st), dev(ac))
            assert np.array_equal(host(kids), kids_o) and np.array_equal(host(done), done_o) and np.array_equal(host(chg), chg_o)
    with pytest.raises(mat_mul_amd.TensorGameError, match="outside"):
        ops.step_many(padded(st), torch.zeros((B, TG_MAX_ACTIONS + 1, 3 * S), dtype=torch.int8, device=DEV))


def test_randomized_differential():
    """Seeded random configurations -- S, batch, game stride padding, byte offset of the buffer,
    vocabulary, shift, action count, mode -- every one compared bit for bit with the oracle."""
    rng = np.random.default_rng(20261004)
    vocabularies = [((0, 1, 2), 1), ((1, 2, 3), 2), ((0, 1, 2, 3, 4), 2), ((-1, 0, 1), 0), (tuple(range(-4, 7)), 1)]
    import os
    total_cases = int(os.environ.get("TG_RANDOM_CASES", "120"))   # raise for a one-off stress run
    n_cases = 0
    for case in range(total_cases):
        S = int(rng.choice([1, 2, 3, 4, 4, 5, 7, 8, 9, 9, 12, 16, 16, 25, 25, 32]))
        B = int(rng.integers(1, 24 if S <= 16 else 6))
        K = int(rng.integers(1, 12))
        toks, shift = vocabularies[int(rng.integers(len(vocabularies)))]
        N = S ** 3
        pad = int(rng.choice([0, 0, 16 - N % 16 if N % 16 else 0, 3, 48]))
        offset = int(rng.choice([0, 0, 0, 16, 5]))
        stride = N + pad
        st = rng.integers(-3, 4, size=(B, S, S, S)).astype(np.int8)
        ac = rng.choice(toks, size=(B, K, 3 * S)).astype(np.int8)
        buf = torch.zeros(B * stride + offset + 64, dtype=torch.int8, device=DEV)
        t = buf[offset:offset + B * stride].view(B, stride)[:, :N].unflatten(1, (S, S, S))
        t.copy_(dev(st))
        mode = case % 4
        ovf = torch.zeros((B,), dtype=torch.uint8, device=DEV)
        if mode == 0:
            want, wd, wo = O.step_i8(st, ac[:, 0], shift)
            out, d = ops.step(t, dev(ac[:, 0]), overflow=ovf, shift=shift)
            ok = np.array_equal(host(out), want) and np.array_equal(host(d), wd) and np.array_equal(host(ovf), wo)
        elif mode == 1:
            want, wds, wo = O.step_many_i8(st, ac, shift)
            out, ds = ops.step_many(t, dev(ac), overflow=ovf, shift=shift)
            ok = np.array_equal(host(out), want) and np.array_equal(host(ds), wds) and np.array_equal(host(ovf), wo)
        elif mode == 2:
            wk, wd, wc, wo = O.expand_i8(st, ac, shift)
            ovk = torch.zeros((B, K), dtype=torch.uint8, device=DEV)
            kids, d, c = ops.expand(t, dev(ac), overflow=ovk, shift=shift)
            ok = (np.array_equal(host(kids), wk) and np.array_equal(host(d), wd) and np.array_equal(host(c), wc)
                  and np.array_equal(host(ovk), wo))
        else:
            want, wo = O.gen_from_factors_i8(ac, shift)
            out = ops.gen_from_factors(dev(ac), S, overflow=ovf, shift=shift)
            ok = np.array_equal(host(out), want) and np.array_equal(host(ovf), wo)
        assert ok, dict(case=case, S=S, B=B, K=K, shift=shift, pad=pad, offset=offset, mode=mode, toks=toks)
        assert np.array_equal(host(t), st) or mode == 3, "inputs must be untouched"
        assert np.array_equal(host(ops.state_hash(t)).view(np.uint64), O.state_hash(st))
        n_cases += 1
    assert n_cases == total_cases


def test_take_action_batched_greedy_step():
    """functional.take_action == the arithmetic of training.py:249-268 restated with the oracle."""
    rng = np.random.default_rng(12)
    S, T, n, groups = 4, 3, 4, 5
    B = n * groups
    st = rng.integers(-1, 2, size=(B, T, S, S, S)).astype(np.int8)
    tok = rng.integers(1, 4, size=(B, 3 * S)).astype(np.int8)          # tokens {1,2,3}, shift 2
    new_state, rank_ubs, best = F.take_action(dev(st), dev(tok), n_samples=n)
    head = st[:, 0].astype(np.int64) - O.action_to_tensor(tok, shift=2)
    want_state = np.concatenate([head[:, None], st[:, :-1].astype(np.int64)], axis=1)
    want_ubs = (head != 0).reshape(B, -1).sum(1).reshape(groups, n)
    assert np.array_equal(host(new_state), want_state)
    assert np.array_equal(host(rank_ubs), want_ubs)
    assert np.array_equal(host(best.values), want_ubs.min(1))
    assert np.array_equal(want_ubs[np.arange(groups), host(best.indices)], want_ubs.min(1))


def test_no_out_of_bounds_writes():
    """Guard bytes before/after every output buffer and in the padding between games must survive every
    entry point, for aligned, padded and packed layouts."""
    GUARD, CANARY = 256, 0x5A
    rng = np.random.default_rng(99)

    def guarded_states(B, S, stride):
        n = S ** 3
        buf = torch.full((GUARD + B * stride + GUARD,), CANARY, dtype=torch.uint8, device=DEV).view(torch.int8)
        view = buf[GUARD:GUARD + B * stride].view(B, stride)[:, :n].unflatten(1, (S, S, S))
        return buf, view

    def check(buf, B, S, stride, what):
        raw = host(buf.view(torch.uint8))
        n = S ** 3
        assert (raw[:GUARD] == CANARY).all() and (raw[GUARD + B * stride:] == CANARY).all(), what
        body = raw[GUARD:GUARD + B * stride].reshape(B, stride)
        assert (body[:, n:] == CANARY).all(), what + " (padding between games)"

    def guarded(shape, dtype):
        numel = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        buf = torch.full((GUARD + numel + GUARD,), CANARY, dtype=torch.uint8, device=DEV)
        return buf, buf[GUARD:GUARD + numel].view(dtype).view(shape)

    def check_flat(buf, what):
        raw = host(buf)
        assert (raw[:GUARD] == CANARY).all() and (raw[-GUARD:] == CANARY).all(), what

    for S in (4, 9, 16, 25, 6):
        n = S ** 3
        for stride in {n, -(-n // 16) * 16, -(-n // 16) * 16 + 32}:
            B, K = 7, 5
            st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
            ac = dev(rng.integers(0, 3, size=(B, K, 3 * S)).astype(np.int8))
            sbuf, src = guarded_states(B, S, stride)
            src.copy_(dev(st))
            for name in ("step", "step_many", "gen_from_factors", "reset_matmul", "reset_broadcast", "change_basis"):
                obuf, out = guarded_states(B, S, stride)
                dbuf, done = guarded((B,), torch.uint8)
                if name == "step":
                    ops.step(src, ac[:, 0].contiguous(), out=out, done=done)
                elif name == "step_many":
                    d2buf, ds = guarded((B,), torch.int32)
                    ops.step_many(src, ac, out=out, done_step=ds)
                    check_flat(d2buf, name)
                elif name == "gen_from_factors":
                    ops.gen_from_factors(ac, S, out=out)
                elif name == "reset_matmul":
                    if int(np.sqrt(S)) ** 2 != S:
                        continue
                    ops.reset_matmul(out, int(np.sqrt(S)))
                elif name == "reset_broadcast":
                    ops.reset_broadcast(out, dev(st[0]))
                else:
                    eye = torch.eye(S, dtype=torch.int32, device=DEV).expand(B, 3, S, S).contiguous()
                    ops.change_basis(src, eye, out=out)
                torch.cuda.synchronize()
                check(obuf, B, S, stride, f"{name} S={S} stride={stride}")
                check_flat(dbuf, name)
            check(sbuf, B, S, stride, f"inputs S={S} stride={stride}")
            # expand: children buffer (B*k games) + flags
            k = 3
            cbuf, kids = guarded_states(B * k, S, stride)
            fbuf, flags = guarded((3, B, k), torch.uint8)
            ops.expand(src, ac[:, :k].contiguous(), out=kids.unflatten(0, (B, k)), done=flags[0], changed=flags[1],
                       overflow=flags[2])
            torch.cuda.synchronize()
            check(cbuf, B * k, S, stride, f"expand S={S} stride={stride}")
            check_flat(fbuf, "expand flags")
        # generator, hash, rank, emit_frames
        tbuf, tok = guarded((5, 4, 3 * S), torch.int8)
        gbuf, tgt = guarded_states(5, S, -(-n // 16) * 16)
        ops.gen_demos(5, S, 4, DEV, seed=1, target=tgt, actions=tok)
        torch.cuda.synchronize()
        check_flat(tbuf, "gen_demos tokens")
        check(gbuf, 5, S, -(-n // 16) * 16, "gen_demos target")
        ring = ops.alloc_ring(3, S, 2, DEV)
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            xbuf, x = guarded((3, 2, S, S, S), dt)
            scbuf, sc = guarded((3, 1), torch.float32)
            if x.data_ptr() % 16 == 0:
                ops.emit_frames(ring, 1, 2.0, dt, out=x, scalars=sc)
                torch.cuda.synchronize()
                check_flat(xbuf, "emit_frames")
                check_flat(scbuf, "emit_frames scalars")


@pytest.mark.parametrize("S,B,R", [(25, 512, 64), (16, 1024, 20), (4, 4096, 7), (25, 4096, 64), (25, 32768, 8)])
def test_cfg5_generate_in_random_basis_then_replay(S, B, R):
    """BASELINE config 5 shape: targets generated in a random unimodular basis; replaying the EMITTED actions
    (any order) takes every game that did not overflow to zero, and the tensor-level change of basis of the
    plain target equals the factor-level one."""
    P = ops.sample_basis(B, S, DEV, seed=31)
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    tok, tgt = ops.gen_demos(B, S, R, DEV, seed=32, basis=P, overflow=ovf)
    ok = ~ovf.bool()
    assert int(ok.sum()) > B // 2
    env = TensorGameEnv(B, S, DEV)
    env.reset(tgt)
    state, done_step = env.step_many(tok.flip(1).contiguous())
    assert not bool(state[ok].any()) and bool((done_step[ok] >= 0).all())
    assert not bool(env.overflow[ok].any())
    plain_tok, plain_tgt = ops.gen_demos(B, S, R, DEV, seed=32)
    ovf2 = torch.zeros(B, dtype=torch.uint8, device=DEV)
    moved = ops.change_basis(plain_tgt, P.to(torch.int32), overflow=ovf2)
    both = ok & ~ovf2.bool()
    assert torch.equal(moved[both], tgt[both])


# ------------------------------------------------------------------ round-2 gap closers
def test_expand_refuses_out_that_cannot_be_addressed_by_one_stride():
    """out = big[:, :k] of a wider (B,k2,...) buffer cannot be addressed as base + (b*k+i)*stride; the
    old code flattened it (a silent copy) and wrote through the original pointer with the copy's strides."""
    B, S, k, k2 = 6, 4, 3, 5
    st, ac = rand_case(np.random.default_rng(3), B, S, k=k)
    big = ops.alloc_states(B * k2, S, DEV).unflatten(0, (B, k2))
    with pytest.raises(mat_mul_amd.TensorGameError, match="evenly spaced"):
        ops.expand(padded(st), dev(ac), out=big[:, :k])
    # evenly spaced views are fine: every second slot of a wider buffer
    wide = ops.alloc_states(B * k * 2, S, DEV)
    out = wide[::2].unflatten(0, (B, k))
    kids, dn, ch = ops.expand(padded(st), dev(ac), out=out)
    wk, wdn, wch, _ = O.expand_i8(st, ac)
    assert np.array_equal(host(kids), wk) and np.array_equal(host(dn), wdn) and np.array_equal(host(ch), wch)
    assert not bool(wide[1::2].any())                       # the slots in between were not touched


def test_sample_basis_refuses_values_that_overflow_int8():
    """P = L @ U is emitted as int8; with |value| = 3 at S = 25 an entry can reach 225 and would wrap."""
    with pytest.raises(mat_mul_amd.TensorGameError, match="S\\*v\\^2"):
        ops.sample_basis(4, 25, DEV, values=(-3, 0, 3), probs=(0.1, 0.8, 0.1))
    import ctypes as C
    thr = O.categorical_thresholds((0.1, 0.8, 0.1))
    val = np.array([-3, 0, 3], np.int8)
    buf = torch.zeros((4, 3, 25, 25), dtype=torch.int8, device=DEV)
    rc = mat_mul_amd._lib.lib.tg_sample_basis_i8(C.c_void_p(buf.data_ptr()), None, None, 4, 25, thr.ctypes.data_as(C.c_void_p),
                                                 val.ctypes.data_as(C.c_void_p), 3, 0, 0, None)
    assert rc == -1 and b"S*v^2" in mat_mul_amd._lib.lib.tg_last_error()
    with pytest.raises(ValueError):
        O.sample_basis(4, 25, thr, (-3, 0, 3), seed=0)
    P, L, U = ops.sample_basis(8, 9, DEV, values=(-3, 0, 3), probs=(0.1, 0.8, 0.1), seed=2, want_factors=True)  # 9*9 <= 127
    Po, Lo, Uo = O.sample_basis(8, 9, thr, (-3, 0, 3), seed=2)
    assert np.array_equal(host(P), Po) and np.array_equal(host(L), Lo) and np.array_equal(host(U), Uo)


@pytest.mark.parametrize("S,B", [(4, 1000), (9, 77), (16, 33), (25, 9), (5, 3), (32, 2)])
def test_copy_states(S, B):
    """tg_copy_i8: every layout pair (padded / packed / byte-offset), padding bytes untouched, env.snapshot()."""
    rng = np.random.default_rng(S)
    st = rng.integers(-128, 128, size=(B, S, S, S)).astype(np.int8)
    n = S ** 3
    src_p, src_k = padded(st), dev(st)
    off = torch.zeros(B * n + 3, dtype=torch.int8, device=DEV)[3:].view(B, S, S, S)  # byte-offset, packed
    off.copy_(src_k)
    for src in (src_p, src_k, off):
        for mk in ("padded", "packed", "wide"):
            if mk == "padded":
                dst = ops.alloc_states(B, S, DEV)
            elif mk == "packed":
                dst = torch.zeros((B, S, S, S), dtype=torch.int8, device=DEV)
            else:
                raw = torch.full((B, n + 48), 77, dtype=torch.int8, device=DEV)
                dst = raw[:, :n].unflatten(1, (S, S, S))
            out = ops.copy_states(src, dst)
            assert out.data_ptr() == dst.data_ptr() and np.array_equal(host(dst), st), (S, mk)
            if mk == "wide":
                assert bool((raw[:, n:] == 77).all())       # padding between games is never written
    env = TensorGameEnv(B, S, DEV)
    env.reset(src_k)
    snap = env.snapshot()
    env.step(dev(rng.integers(0, 3, size=(B, 3 * S)).astype(np.int8)))
    assert np.array_equal(host(snap), st)                    # the snapshot is not a view of the live state
    assert ops.copy_states(ops.alloc_states(0, S, DEV)).shape[0] == 0


def test_demo_io_round_trip_from_gpu(tmp_path):
    """N4 on the GPU: demos generated on the device -> SyntheticDemos.save / load_packed, and
    export_reference_layout / import_reference_layout (the reference's per-demo files, datasets.py:62-69) ->
    the bytes come back, and replaying the re-imported actions on the re-imported targets reaches zero."""
    from mat_mul_amd import demo_io
    for S, B, R in [(4, 300, 7), (16, 40, 20), (25, 12, 64)]:
        demos = SyntheticDemos(R, B, 1, S, DEV, seed=100 + S, game_id_offset=17)
        f = tmp_path / f"demos_{S}.tgd"
        demos.save(f)
        tok, tgt, meta = demo_io.load_packed(f, device=DEV)
        assert meta == {"B": B, "R": R, "S": S, "shift": 1, "seed": 100 + S, "game_id_offset": 17}
        assert torch.equal(tok, demos.action_seq) and torch.equal(tgt, demos.target_tensor)
        d = tmp_path / f"ref_{S}"
        assert demos.export_reference_layout(d) == B
        seq0 = torch.load(d / "action_seq_17.pt")            # the reference's layout: list of R int64 (3S,) + fp32 (S,S,S)
        t0 = torch.load(d / "target_tensor_17.pt")
        assert isinstance(seq0, list) and len(seq0) == R and seq0[0].dtype == torch.int64 and tuple(seq0[0].shape) == (3 * S,)
        assert t0.dtype == torch.float32 and tuple(t0.shape) == (S, S, S)
        tok2, tgt2 = demo_io.import_reference_layout(d, B, start_index=17)
        assert torch.equal(tok2.to(DEV), demos.action_seq) and torch.equal(tgt2.to(DEV), demos.target_tensor)
        final, done_step = ops.step_many(padded(tgt2.numpy()), tok2.to(DEV))
        assert not bool(final.any()) and bool((done_step >= 0).all())


def test_integration_md_stub_runs_against_strassen_golden(golden, tmp_path):
    """INTEGRATION.md's ctypes stub -- the reference-side binding for act.py:183 / utils.py:181 -- extracted
    verbatim from the document and executed: the Strassen replay must reproduce the recorded reference states."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    md = (root / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(.*?)```", md, flags=re.S).group(1)
    assert "tensor_game_ffi.py" in code and "def step(" in code
    code = code.replace('C.CDLL("libtensorgame.so")', f'C.CDLL({str(mat_mul_amd._lib.LIB_PATH)!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = golden("strassen")
    state = dev(g["replay"][0][None].astype(np.int8)).contiguous()
    done = torch.zeros(1, dtype=torch.uint8, device=DEV)
    for k in range(7):
        ns["step"](state, dev(g["tokens"][k][None].astype(np.int8)), done)
        torch.cuda.synchronize()
        assert np.array_equal(host(state)[0], g["replay"][k + 1]) and int(done[0]) == int(g["done"][k + 1])
    # the stub's expand binding: children of the start state for the first two Strassen actions
    lib = ns["lib"]
    par = dev(g["replay"][0][None].astype(np.int8)).contiguous()
    acts = dev(g["tokens"][:2][None].astype(np.int8)).contiguous()
    kids = torch.zeros((2, 4, 4, 4), dtype=torch.int8, device=DEV)
    dn = torch.zeros(2, dtype=torch.uint8, device=DEV)
    ch = torch.zeros(2, dtype=torch.uint8, device=DEV)
    ns["_check"](lib.tg_expand_i8(par.data_ptr(), kids.data_ptr(), acts.data_ptr(), dn.data_ptr(), ch.data_ptr(), None,
                                  1, 4, 2, 64, 64, 1, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    wk, wdn, wch, _ = O.expand_i8(g["replay"][0][None].astype(np.int8), g["tokens"][:2][None].astype(np.int8))
    assert np.array_equal(host(kids), wk[0]) and np.array_equal(host(dn), wdn[0]) and np.array_equal(host(ch), wch[0])


def test_bench_two_ranks_self_launched_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher: bench.py starts its own ranks (torch.distributed.run children).
    On the one-GPU box both ranks share the GPU and the control plane is gloo (TG_BENCH_BACKEND).  The contract line is
    the LAST line of stdout, under 4 KB, and carries the sharded extras of an N>1 run: S=16 weak (value_s16) and strong,
    config 4 strong with its single-GPU denominator, the resident stepper on the share, a short cpu_baseline."""
    import json, os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TG_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith("{") and len(lines[-1]) < 4096
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 20 and out["samples"] == 9
    assert out["config"]["global_batch"] == 131072 and out["config"]["batch_per_gpu"] == 65536
    assert out["value"] > 1e9 and 0 < out["roofline"]["frac"] <= 1.0
    assert out["value_s16"] > 1e7 and out["roofline_s16"]["kernel"].startswith("tg::s16_step_kernel") and out["ms_per_step_s16"] > 0
    assert out["s16_strong"]["global_batch"] == 8192 and out["s16_strong"]["batch_per_gpu"] == 4096 and out["s16_strong"]["value"] > 1e7
    s4 = out["s4_strong"]
    assert s4["global_batch"] == 1 << 20 and s4["batch_per_gpu"] == 1 << 19 and s4["ideal"] == 2
    assert s4["one_gpu_launch_us"] > 0 and s4["speedup_event"] > 0 and s4["speedup_wall"] > 0
    st = out["streamed_s4"]
    assert st["ok"] and st["share_us_per_step"] > 0 and st["one_gpu_us_per_step"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    # --scaling strong: config 4 is the headline and the weak figure moves to the side
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--scaling", "strong",
                          "--global-batch", "131072", "--no-also", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 131072 and out["config"]["batch_per_gpu"] == 65536


# ------------------------------------------------------------------ alternating sweep direction of tg_step_i8
@pytest.mark.parametrize("S,B", [(4, 270001), (16, 1031), (16, 5), (25, 37), (25, 1), (25, 9)])
def test_step_is_the_same_in_both_sweep_directions(S, B):
    """Round 3: consecutive tg_step_i8 launches take the games in alternating order (the tail of one sweep is the head of
    the next: L2 / Infinity Cache hits); the workgroup -> game map is a permutation for every grid size, ragged last
    workgroups included.  Four launches in a row (both directions, twice), out of place and in place."""
    rng = np.random.default_rng(S * 7 + B)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
    st[::3] = O.action_to_tensor(ac[::3]).astype(np.int8)
    want, want_done, _ = O.step_i8(st, ac)
    src, acd = dev(st), dev(ac)
    for _ in range(4):
        out, done = ops.step(src, acd)
        assert np.array_equal(host(out), want) and np.array_equal(host(done), want_done)
    for _ in range(2):
        t = padded(st)
        _, done = ops.step(t, acd, out=t)
        assert np.array_equal(host(t), want) and np.array_equal(host(done), want_done)


# ------------------------------------------------------------------ tg_step_stream_i8 (K steps, one launch, actions arriving step by step)
@pytest.mark.parametrize("S,B,K", [(4, 1, 3), (4, 16, 5), (4, 70, 9), (4, 1000, 14), (4, 4099, 6),
                                   (16, 1, 4), (16, 7, 9), (16, 130, 6), (16, 1030, 5),
                                   (25, 1, 4), (25, 6, 11), (25, 37, 9), (25, 210, 5),
                                   (16, 8300, 3), (25, 4200, 3)])     # (the last two: more games than stay resident -- rounds)
def test_step_stream_equals_k_single_steps(S, B, K):
    """Every step of the streamed stepper equals tg_step_i8 / the oracle: state, done[k], sticky overflow; ragged
    batches, terminal games, an overflowing game, progress words; padded and packed layouts."""
    rng = np.random.default_rng(B * 31 + K + S)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    st[::5] = O.action_to_tensor(ac[0, ::5]).astype(np.int8)              # these are done after step 0
    if B > 3:
        st[3] = 127
        ac[1, 3] = 0                                                       # factors -1: 127 - (-1) overflows at step 1
        ac[2, 2] = rng.integers(-3, 6, size=3 * S)                         # wide factors (int16 path limits)
        ac[K - 1, 1::4] = rng.integers(0, 3, size=ac[K - 1, 1::4].shape)   # dense actions (S=16: more candidate rows than the queue holds)
    want_done, want_ovf, cur = np.zeros((K, B), np.uint8), np.zeros(B, np.uint8), st.copy()
    for k in range(K):
        cur, d, o = O.step_i8(cur, ac[k])
        want_done[k] = d
        want_ovf |= o
    n_units, gpu = ops.step_stream_layout(B, S, DEV)
    assert n_units * gpu >= B and (n_units - 1) * gpu < B
    # (S = 25: the packed layout has a 15 625-byte stride; the streamed stepper takes 16-byte-multiple strides only)
    for t in ((padded(st),) if S == 25 else (padded(st), dev(st))):
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        out, done = ops.step_stream(t, dev(ac), overflow=ovf, progress=prog, status=status)
        torch.cuda.synchronize()
        assert out.data_ptr() == t.data_ptr()
        assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf)
        assert bool((prog == K).all()) and int(status[0]) == 0
    assert want_done[0, ::5].all() and (B <= 3 or want_ovf[3] == 1)


def test_step_stream_waits_for_ready_words_and_times_out():
    """The stepper polls ready[k]: released one by one from a second stream while the kernel is resident, every step
    still equals the oracle; with a ready word that never arrives the bounded spin gives up and sets status."""
    rng = np.random.default_rng(8)
    B, K, S = 300, 6, 4
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    cur = st.copy()
    for k in range(K):
        cur, _, _ = O.step_i8(cur, ac[k])
    t = padded(st)
    ready = torch.zeros(K, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    n_units, _ = ops.step_stream_layout(B, S, DEV)
    prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
    acd = dev(ac)
    torch.cuda.synchronize()
    # the producer runs on a HIGH-priority stream: HIP keeps separate hardware queues per priority, so its fill kernels
    # never queue up behind the resident stepper (two streams of one priority may share a queue)
    side, prod = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(side):
        ops.step_stream(t, acd, ready=ready, progress=prog, status=status)
    with torch.cuda.stream(prod):
        for k in range(K):
            ready[k:k + 1].fill_(1)
            prod.synchronize()
    side.synchronize()
    if int(status[0]) == 1:
        pytest.fail("the stepper gave up waiting (status 1) although the producer released every step within milliseconds on a high-priority stream")
    assert np.array_equal(host(t), cur) and bool((prog == K).all())
    # never released: every wavefront gives up after its bounded spin, the state stops where the words stopped
    t2 = padded(st)
    ready.zero_()
    ready[:2] = 1
    status.zero_()
    ops.step_stream(t2, acd, ready=ready, status=status)
    torch.cuda.synchronize()
    two = st.copy()
    for k in range(2):
        two, _, _ = O.step_i8(two, ac[k])
    assert int(status[0]) == 1 and np.array_equal(host(t2), two)


def test_step_stream_beyond_the_resident_batch_runs_in_rounds_without_ready_words():
    """S=4 with more games than the device keeps resident: refused with ready words (a producer waiting for the whole batch
    would stall), accepted without -- units of 64 games run in rounds; BASELINE config 4 whole on one GPU."""
    B, K = 1 << 20, 3
    with pytest.raises(mat_mul_amd.TensorGameError, match="resident"):
        ops.step_stream_layout(B, 4, DEV)
    tok, tgt = ops.gen_demos(B, 4, K, DEV, seed=5)
    st = ops.alloc_states(B, 4, DEV)
    st.copy_(tgt)
    acts = tok.permute(1, 0, 2).contiguous()
    with pytest.raises(mat_mul_amd.TensorGameError, match="resident"):
        ops.step_stream(st, acts, ready=torch.ones(K, dtype=torch.int32, device=DEV))
    prog = torch.zeros(B // 64, dtype=torch.int32, device=DEV)          # in rounds: units of 64 games
    _, done = ops.step_stream(st, acts, progress=prog)
    ref = ops.alloc_states(B, 4, DEV)
    ref.copy_(tgt)
    for k in range(K):
        _, d = ops.step(ref, acts[k], out=ref)
        assert torch.equal(done[k], d)
    assert torch.equal(st, ref) and not bool(st.any()) and bool((prog == K).all())


@pytest.mark.parametrize("S", [16, 25])
def test_step_stream_capacity_and_ready_words_beyond_it(S):
    """at S=16 / 25 tg_step_stream_layout accepts any B and the units run in rounds beyond the resident batch --
    with ready words a producer that waits for the whole batch would never see the later rounds start.  The capacity is
    now exported (from the occupancy of the stepper's kernel on this device); ready words beyond it are refused, the same
    batch without them runs in rounds and equals K single steps."""
    cap = ops.step_stream_capacity(S, DEV)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert cap == cus * {16: 32, 25: 16}[S]
    assert ops.step_stream_capacity(4, DEV) >= 131072      # BASELINE config 4's share of 8 GPUs stays resident
    B, K = cap + 37, 3
    tok, tgt = ops.gen_demos(B, S, K, DEV, seed=9)
    st = ops.alloc_states(B, S, DEV)
    st.copy_(tgt)
    acts = tok.permute(1, 0, 2).contiguous()
    with pytest.raises(mat_mul_amd.TensorGameError, match="resident"):
        ops.step_stream(st, acts, ready=torch.ones(K, dtype=torch.int32, device=DEV))
    assert torch.equal(st, tgt)                          # refused before anything ran
    prog = torch.zeros(B, dtype=torch.int32, device=DEV)
    _, done = ops.step_stream(st, acts, progress=prog)
    ref = ops.alloc_states(B, S, DEV)
    ref.copy_(tgt)
    for k in range(K):
        _, d = ops.step(ref, acts[k], out=ref)
        assert torch.equal(done[k], d)
    assert torch.equal(st, ref) and not bool(st.any()) and bool((prog == K).all())
    # at the capacity itself ready words are fine
    st2 = ops.alloc_states(cap, S, DEV)
    st2.copy_(tgt[:cap])
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.step_stream(st2, acts[:, :cap].contiguous(), ready=torch.ones(K, dtype=torch.int32, device=DEV), status=status)
    assert int(status[0]) == 0 and not bool(st2.any())


@pytest.mark.parametrize("S", [4, 16, 25])
def test_env_step_stream_replays_demonstrations_to_zero(S):
    """TensorGameEnv.step_stream: R generated actions per game in one resident launch bring every target to zero, done[k]
    as the per-step env reports it."""
    B, R = 50, 6
    tok, tgt = ops.gen_demos(B, S, R, DEV, seed=S)
    env = mat_mul_amd.TensorGameEnv(B, S, device=DEV)
    env.reset(tgt)
    ref = mat_mul_amd.TensorGameEnv(B, S, device=DEV)
    ref.reset(tgt)
    acts = tok.permute(1, 0, 2).contiguous()
    state, done = env.step_stream(acts)
    torch.cuda.synchronize()
    for k in range(R):
        _, d = ref.step(acts[k])
        assert torch.equal(done[k], d)
    assert torch.equal(state, ref.state) and not bool(state.any()) and bool(env.done.all()) and env.t == R


@pytest.mark.parametrize("S,B,K", [(4, 300, 21), (4, 5000, 11), (16, 70, 19), (25, 26, 19)])
def test_step_stream_takes_released_steps_in_blocks(S, B, K):
    """Round 3: a wavefront takes all the steps it finds released at once (up to 8).  Ready words pre-set with GAPS
    (a set word behind an unset one must not be taken), the rest released in bursts of 1..9 from a second stream while the
    stepper is resident: state, every done[k], progress and status as K single steps."""
    rng = np.random.default_rng(S * 1000 + B + K)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    want_done, cur = np.zeros((K, B), np.uint8), st.copy()
    for k in range(K):
        cur, want_done[k], _ = O.step_i8(cur, ac[k])
    t, acd = padded(st), dev(ac)
    ready = torch.zeros(K, dtype=torch.int32, device=DEV)
    ready[:3] = 1
    ready[4:6] = 1                                                        # behind the gap at 3: not to be taken yet
    ready[K - 1] = 1
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    n_units, _ = ops.step_stream_layout(B, S, DEV)
    prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
    done = torch.zeros((K, B), dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()
    side, prod = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)     # (separate hardware queues: see the tests above)
    with torch.cuda.stream(side):
        ops.step_stream(t, acd, done=done, ready=ready, progress=prog, status=status)
    with torch.cuda.stream(prod):
        k = 3
        for burst in (1, 4, 2, 9, 1, 3, 8, 8):
            ready[k:min(K, k + burst)].fill_(1)
            prod.synchronize()
            k += burst
            if k >= K:
                break
    side.synchronize()
    if int(status[0]) == 1:
        pytest.fail("the stepper gave up waiting (status 1) although the producer released every step within milliseconds on a high-priority stream")
    assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done) and bool((prog == K).all())


@pytest.mark.parametrize("S,B", [(4, 700), (16, 90), (25, 30)])
def test_step_stream_hand_off_is_visible_while_the_kernel_runs(S, B):
    """The publish protocol end to end: a consumer (this test, on another stream) waits for progress[u] >= k on every
    unit, READS the state while the stepper is still resident -- it must already equal the oracle's state after k steps
    (write-through stores drained before the progress word) -- and only then releases action block k."""
    rng = np.random.default_rng(S + B)
    K = 5
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    t = padded(st)
    acd = dev(ac)
    ready = torch.zeros(K, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    n_units, _ = ops.step_stream_layout(B, S, DEV)
    prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
    done = torch.zeros((K, B), dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()
    # The stepper runs on a normal-priority stream, the consumer on a HIGH-priority one: HIP keeps separate hardware
    # queues per priority, so the consumer's small kernels never queue up behind the resident stepper (two streams of
    # the same priority may share a hardware queue -- then nothing of the consumer runs until the stepper's bounded
    # wait expires).
    side, cons = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(side):
        ops.step_stream(t, acd, done=done, ready=ready, progress=prog, status=status)
    import time
    cur = st.copy()
    with torch.cuda.stream(cons):
        for k in range(K):
            t0 = time.time()
            while int(prog.min()) < k:                                   # (a tiny kernel + copy on the consumer stream)
                if time.time() - t0 > 20:
                    side.synchronize()
                    if int(status[0]) == 1:
                        pytest.fail("the stepper gave up waiting (status 1) although the consumer released every step on a high-priority stream")
                    raise AssertionError("the stepper made no progress")
            seen = host(t.clone())                                        # read while the stepper is resident
            assert np.array_equal(seen, cur), (S, k)
            if k:
                assert np.array_equal(host(done[k - 1].clone()), want_done), (S, k)
            cur, want_done, _ = O.step_i8(cur, ac[k])
            ready[k:k + 1].fill_(1)
            cons.synchronize()
    side.synchronize()
    assert int(status[0]) == 0 and np.array_equal(host(t), cur) and bool((prog == K).all())


def test_step_stream_refuses_what_it_does_not_implement():
    t = ops.alloc_states(8, 9, DEV)
    with pytest.raises(mat_mul_amd.TensorGameError, match="S=4, S=16 and S=25"):
        ops.step_stream(t, torch.ones((2, 8, 27), dtype=torch.int8, device=DEV))
    assert ops.step_stream_layout(8192, 16, DEV) == (8192, 1)


def test_step_stream_layout_never_exceeds_what_the_device_keeps_resident():
    """the S=4 layout must come from the occupancy of the variant it selects: every accepted batch has all its
    units resident at once, larger batches are refused instead of stalling a producer that waits for the whole batch.
    Round 4: from 57 344 games on the one-game-per-lane kernel (64 games per unit) takes every batch it holds."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    cap = ops.step_stream_capacity(4, DEV)
    seen = set()
    for B in [1, 15, 16, 17, 4096, 57343, 57344, 65536, 131072, 200000, 262144, 300000, 400000, 458752, 500000, 1 << 19, 1 << 20]:
        try:
            units, gpu = ops.step_stream_layout(B, 4, DEV)
        except mat_mul_amd.TensorGameError as e:
            assert "resident" in str(e) and B > 131072 and B > cap
            continue
        assert B <= cap
        assert gpu in (16, 32, 64) and units == -(-B // gpu)
        assert units <= cus * 32                                          # never more than 8 wavefronts per SIMD
        if 57344 <= B <= 262144:
            assert gpu == 64 and units <= cus * 16                        # the lane kernel: four wavefronts per SIMD
        if B < 57344:
            assert gpu == 16
        seen.add(gpu)
    assert {16, 64}.issubset(seen) and cap >= 262144
    # a batch the lane kernel takes, ragged in its last unit, still steps exactly, all units reporting
    B, K = 200001, 3
    units, gpu = ops.step_stream_layout(B, 4, DEV)
    tok, tgt = ops.gen_demos(B, 4, K, DEV, seed=3)
    st = ops.alloc_states(B, 4, DEV)
    st.copy_(tgt)
    prog = torch.zeros(units, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.step_stream(st, tok.permute(1, 0, 2).contiguous(), progress=prog, status=status)
    assert not bool(st.any()) and bool((prog == K).all()) and int(status[0]) == 0


@pytest.mark.parametrize("B,K", [(57344, 5), (65536 + 37, 19), (131072, 9)])
def test_step_stream_one_game_per_lane_equals_k_single_steps(B, K):
    """Round 4: s4_stream_kernel_lanes (a lane owns a game, 64 games per wavefront, tokens double-buffered, the state leaving
    transposed through LDS): every step equals the oracle -- state, done[k], sticky overflow -- with terminal games, an
    overflowing game, wide factors (the whole wavefront then takes the general form on its LDS image), dense actions, a
    ragged last unit, more steps than two blocks, ready words pre-set and absent."""
    S = 4
    rng = np.random.default_rng(B + K)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    st[1000:1064] = rng.integers(-120, 121, size=(64, S, S, S))            # large entries: L1 norm beyond the digit form's limit
    st[::5] = O.action_to_tensor(ac[0, ::5]).astype(np.int8)              # these are done after step 0
    st[3] = 127
    ac[1, 3] = 0                                                           # factors -1: 127 - (-1) overflows at step 1
    ac[2, 2] = rng.integers(-3, 6, size=3 * S)                             # wide factors: the general form, wavefront 0
    ac[K - 1, 1::4] = rng.integers(0, 3, size=ac[K - 1, 1::4].shape)       # dense actions
    assert (B - 1) % 5
    ac[:, B - 1] = rng.integers(-128, 128, size=(K, 3 * S))                # the last game: full-range tokens at every step
    want_done, want_ovf, cur = np.zeros((K, B), np.uint8), np.zeros(B, np.uint8), st.copy()
    for k in range(K):
        cur, d, o = O.step_i8(cur, ac[k])
        want_done[k] = d
        want_ovf |= o
    n_units, gpu = ops.step_stream_layout(B, S, DEV)
    assert gpu == 64 and n_units == -(-B // 64)
    acd = dev(ac)
    for ready in (None, torch.ones(K, dtype=torch.int32, device=DEV)):
        t = padded(st)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        out, done = ops.step_stream(t, acd, overflow=ovf, ready=ready, progress=prog, status=status)
        torch.cuda.synchronize()
        assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf)
        assert bool((prog == K).all()) and int(status[0]) == 0
    assert want_done[0, ::5].all() and want_ovf[3] == 1


def test_step_stream_one_game_per_lane_with_a_producer_in_bursts():
    """The lane kernel's protocol: ready words pre-set with GAPS, the rest released in bursts from a high-priority stream
    while 65 536 games are resident; a consumer that waits for progress sees the state of every published step."""
    S, B, K = 4, 65536, 27
    rng = np.random.default_rng(77)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    want_done, cur = np.zeros((K, B), np.uint8), st.copy()
    for k in range(K):
        cur, want_done[k], _ = O.step_i8(cur, ac[k])
    t, acd = padded(st), dev(ac)
    ready = torch.zeros(K, dtype=torch.int32, device=DEV)
    ready[:3] = 1
    ready[4:6] = 1                                                        # behind the gap at 3: not to be taken yet
    ready[K - 1] = 1
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    n_units, gpu = ops.step_stream_layout(B, S, DEV)
    assert gpu == 64
    prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
    done = torch.zeros((K, B), dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()
    side, prod = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(side):
        ops.step_stream(t, acd, done=done, ready=ready, progress=prog, status=status)
    with torch.cuda.stream(prod):
        k = 3
        for burst in (1, 4, 2, 9, 1, 3, 8, 8):
            ready[k:min(K, k + burst)].fill_(1)
            prod.synchronize()
            k += burst
            if k >= K:
                break
    side.synchronize()
    assert int(status[0]) == 0, "the stepper gave up waiting although every step was released"
    assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done) and bool((prog == K).all())


# ------------------------------------------------------------------ change of basis on the matrix cores
@pytest.mark.parametrize("S,B", [(9, 37), (16, 21), (25, 9)])
def test_change_basis_matrix_core_kernel_and_its_fallbacks(S, B):
    """tg_change_basis_i8 (A12; parity unpinned: the oracle is the paper-level einsum): sparse unimodular bases and their
    exact inverses, dense {-1,0,1} matrices, entries at +-127, entries beyond int8 (vector form inside the same launch),
    targets that overflow int8 (wrap + flag), the state -128, identity."""
    rng = np.random.default_rng(S * 7 + B)
    thr = O.categorical_thresholds((0.15, 0.7, 0.15))
    _, tgt, _ = O.gen_demos_i8(B, S, 20, thr, (-1, 0, 1), 1, seed=S)
    Pm, L, U = O.sample_basis(B, S, O.categorical_thresholds((0.03, 0.94, 0.03)), (-1, 0, 1), seed=S + 1)
    bases = {"sparse unimodular": Pm, "its inverse": O.unimodular_inverse(L, U),
             "dense ternary": rng.integers(-1, 2, size=(B, 3, S, S)),
             "entries up to 127": np.where(rng.random((B, 3, S, S)) < 0.04, rng.integers(-127, 128, size=(B, 3, S, S)), 0) + np.eye(S, dtype=np.int64),
             "entries beyond int8": np.where(rng.random((B, 3, S, S)) < 0.03, rng.integers(-1000, 1001, size=(B, 3, S, S)), 0) + np.eye(S, dtype=np.int64),
             "identity": np.broadcast_to(np.eye(S, dtype=np.int64), (B, 3, S, S)).copy()}
    states = {"demo targets": tgt, "full range": rng.integers(-128, 128, size=(B, S, S, S)).astype(np.int8)}
    for sname, st in states.items():
        for bname, M in bases.items():
            want, want_ovf = O.change_basis_i8(st, M)
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            got = ops.change_basis(padded(st), dev(M.astype(np.int32)), overflow=ovf)
            assert np.array_equal(host(got), want), (S, sname, bname)
            assert np.array_equal(host(ovf), want_ovf), (S, sname, bname)
    want, want_ovf = O.change_basis_i8(tgt, Pm)
    assert (want_ovf == 0).sum() > 0                        # the sparse basis keeps some targets in range
    # intermediates in [32640, 32767] do not fit the two int8 byte planes of the matrix-core path (the high
    # plane would hold 128); such games must take the vector form.  x = -128 along k, a row of C summing to -255 (and
    # 254 / 256 either side of the edge), then the same through mode 2.
    for tot in (254, 255, 256):
        M = np.broadcast_to(np.eye(S, dtype=np.int64), (B, 3, S, S)).copy()
        M[:, 2, 0, :3] = [-127, -(tot - 128), -1]
        st = rng.integers(-1, 2, size=(B, S, S, S)).astype(np.int8)
        st[:, :, :, :3] = -128
        M2 = M.copy()
        M2[:, 1, 1, :2] = [1, 1]                             # mode 2 doubles what mode 3 produced
        for MM in (M, M2):
            want, want_ovf = O.change_basis_i8(st, MM)
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            got = ops.change_basis(padded(st), dev(MM.astype(np.int32)), overflow=ovf)
            assert np.array_equal(host(got), want) and np.array_equal(host(ovf), want_ovf), (S, tot)


# ------------------------------------------------------------------ outputs beyond the cache
@pytest.mark.parametrize("S,B,k", [(4, 262144, 8), (9, 23000, 8), (16, 4100, 8), (25, 1100, 8)])
def test_streamed_outputs_match_the_cached_form(S, B, k):
    """Outputs of >= 128 MiB (expand's children -- S = 4 writes them with non-temporal stores from that size on --,
    copies, resets, model-input frames): the same bytes as a small slice of the batch gives, and (size-independent
    property) child c of a parent == one step of the parent with action c."""
    tok, tgt = ops.gen_demos(B, S, k, DEV, seed=S + 1)
    tok[1, k - 1, :S] = 1                                      # a null action
    kids, done, chg = ops.expand(tgt, tok)
    assert kids.numel() >= 128 << 20 or S == 9                 # (S=9: 736-byte stride, 135 MB with padding)
    n = 64
    kids_small, done_small, chg_small = ops.expand(tgt[:n], tok[:n].contiguous())
    assert torch.equal(kids[:n], kids_small) and torch.equal(done[:n], done_small) and torch.equal(chg[:n], chg_small)
    for c in (0, k - 1):
        st, dn = ops.step(tgt, tok[:, c].contiguous())
        assert torch.equal(kids[:, c], st) and torch.equal(done[:, c], dn)
    assert int(chg[1, k - 1]) == 0 and int(chg.sum()) < chg.numel()
    # copies and resets of the children's buffer (same footprint)
    flat = kids.flatten(0, 1)
    cp = ops.copy_states(flat)
    assert torch.equal(cp, flat)
    ops.reset_broadcast(cp, tgt[0].contiguous())
    assert bool((cp == tgt[0]).all())
    del cp, kids
    # model-input frames: float32 (B, T, S, S, S) of >= 128 MiB
    T = max(2, -(-(128 << 20) // (B * S ** 3 * 4)))
    ring = torch.randint(-3, 4, (B, T, S, S, S), dtype=torch.int8, device=DEV)
    x, _ = ops.emit_frames(ring, T - 1, 0.0)
    assert x.numel() * 4 >= 128 << 20
    assert torch.equal(x, ring.flip(1).float())
