"""NumPy restatement of the reference's tensor-game arithmetic.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a CPU restatement of the hot path of kurtosis/mat_mul
(``/root/reference``), written from the reference's behaviour, each function citing
the reference ``file:line`` it follows.  It is pinned against the reference itself by
``tests/golden/*.npz`` (written by ``tests/golden/make_golden.py``, which imports the
reference in the build container) -- see ``tests/test_oracle_golden.py``.

It is NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` import it, and only as the checker.  The product
(``mat_mul_amd``) never imports it and has no CPU fallback.

Two layers:

* "reference-semantics" functions (``action_to_uvw`` ... ``build_matmul_tensor``):
  the reference's functions on integer arrays, arithmetic in int64 (the reference
  computes in float32/int64 on small integers, which is exact, so integers agree).
* "build-semantics" functions (``step_i8`` ... ``gen_demos_i8``): what the HIP
  kernels behind ``include/tensor_game.h`` must return, bit for bit, for int8
  states / int8 tokens, including the wrap-around + sticky overflow flag and the
  Philox-4x32-10 generator stream.  These are defined by composing the layer above.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# Layer 1: the reference's functions, restated on integer ndarrays
# --------------------------------------------------------------------------------------


def action_to_uvw(action, shift: int = 1):
    """reference utils.py:56-66 -- ``(action - shift).split(dim_3d, dim=-1)``."""
    a = np.asarray(action).astype(np.int64) - int(shift)
    S = a.shape[-1] // 3
    return a[..., :S], a[..., S : 2 * S], a[..., 2 * S : 3 * S]


def uvw_to_tensor(uvw):
    """reference utils.py:69-85 -- rank-1 outer product, element [i,j,l] = u_i v_j w_l
    (1-D branch :77-78 and batched broadcasting branch :80-84 are the same formula)."""
    u, v, w = (np.asarray(x).astype(np.int64) for x in uvw)
    return u[..., :, None, None] * v[..., None, :, None] * w[..., None, None, :]


def action_to_tensor(action, shift: int = 1):
    """reference utils.py:88-96 -- uvw_to_tensor(action_to_uvw(action))."""
    return uvw_to_tensor(action_to_uvw(action, shift))


def get_head_state(state, unsqueeze: bool = True):
    """reference utils.py:99-111 -- frame 0 of (B,T,S,S,S)."""
    state = np.asarray(state)
    return state[:, 0:1] if unsqueeze else state[:, 0]


def tensor_factorized(state) -> bool:
    """reference utils.py:181-188 -- VERBATIM semantics ``(state[0] == 0).all()``:
    inspects index 0 of the leading axis only."""
    return bool((np.asarray(state)[0] == 0).all())


def done_per_game(head) -> np.ndarray:
    """The intended batched terminal test: head of game b is all zero.  Equal to
    ``tensor_factorized(get_head_state(state[b:b+1]))`` per game (utils.py:181-188
    as called at act.py:177)."""
    head = np.asarray(head)
    return (head.reshape(head.shape[0], -1) == 0).all(axis=1)


def get_child_states(state, actions, shift: int = 1):
    """reference act.py:266-275 -- the env step for k candidate actions.
    state (B,T,S,S,S), actions (B,k,3S) -> list of k arrays (B,T,S,S,S):
    new head = head - action tensor (:268-270); history shifts right, oldest frame
    dropped (:271-274).  Inputs untouched."""
    state = np.asarray(state).astype(np.int64)
    actions = np.asarray(actions)
    k = actions.shape[1]
    action_tensor = action_to_tensor(actions, shift)  # (B,k,S,S,S)
    new_heads = get_head_state(state) - action_tensor  # (B,1,...) - (B,k,...)
    return [
        np.concatenate([new_heads[:, i : i + 1], state[:, :-1]], axis=1) for i in range(k)
    ]


def take_actions(action_seq, target, shift: int = 1):
    """reference datasets.py:144-153 -- sequential ``target - action_to_tensor(a)``."""
    t = np.asarray(target).astype(np.int64)
    for a in action_seq:
        t = t - action_to_tensor(a, shift)
    return t


def remove_null_actions(state, candidate_states):
    """reference utils.py:191-194 -- indexes of candidates whose head differs from the parent."""
    state = np.asarray(state)
    return [i for i, c in enumerate(candidate_states) if (np.asarray(c)[:, 0] != state[:, 0]).any()]


def build_matmul_tensor(dim_t: int, dim_i: int, dim_j: int, dim_k: int):
    """reference utils.py:143-161 -- <n,n,n> matmul tensor in frame 0.  The reference's
    index formula (:158-160) is only meaningful for square shapes;
    like the build, the oracle accepts square shapes only."""
    if not (dim_i == dim_j == dim_k):
        raise ValueError("only square matmul tensors are defined (reference utils.py:160 mixes dim_j/dim_k)")
    n = dim_i
    t = np.zeros((dim_t, n * n, n * n, n * n), dtype=np.int64)
    for ik in range(n * n):
        for j in range(n):
            t[0, (ik // n) * n + j, j * n + ik % n, ik] = 1
    return t


def uvw_to_demo(uu, vv, ww, shift: int = 1):
    """reference utils.py:40-53 -- sum_i u_i (x) v_i (x) w_i and the token table
    ``cat(uu,vv,ww)+shift`` (the reference hard-codes 4x4x4 at :45; generalised to S)."""
    uu, vv, ww = (np.asarray(x).astype(np.int64) for x in (uu, vv, ww))
    tensor = uvw_to_tensor((uu, vv, ww)).sum(axis=0)
    return tensor, np.concatenate([uu, vv, ww], axis=1) + int(shift)


def nnz_per_game(head) -> np.ndarray:
    """reference training.py:266 -- ``sum(head != 0)`` per game (rank upper bound)."""
    head = np.asarray(head)
    return (head.reshape(head.shape[0], -1) != 0).sum(axis=1).astype(np.int32)


def demo_getitem(action_seq, target, idx_action: int, dim_t: int, shift: int = 1):
    """reference datasets.py:84-122 (the arithmetic of SyntheticDemoDataset.__getitem__,
    file I/O excluded): state frames (dim_t,S,S,S), scalar, action, reward."""
    R = len(action_seq)
    t = np.asarray(target).astype(np.int64)
    if idx_action != R - 1:
        t = take_actions(action_seq[idx_action + 1 :], t, shift)
    frames = [t] + [
        action_to_tensor(a, shift) for a in reversed(action_seq[idx_action + 1 : idx_action + dim_t])
    ]
    frames = np.stack(frames)
    if len(frames) < dim_t:
        frames = np.concatenate([frames, np.zeros((dim_t - len(frames),) + frames.shape[1:], np.int64)])
    return frames, float(R - idx_action), np.asarray(action_seq[idx_action]), float(-(idx_action + 1))


# --------------------------------------------------------------------------------------
# Layer 2: build semantics (int8 states, int8 tokens) -- what the HIP kernels must return
# --------------------------------------------------------------------------------------


def _narrow_i8(x64):
    """int64 -> (int8 with two's-complement wrap, per-game overflow flag)."""
    x64 = np.asarray(x64)
    B = x64.shape[0]
    ovf = ((x64 < -128) | (x64 > 127)).reshape(B, -1).any(axis=1).astype(np.uint8)
    return x64.astype(np.int8), ovf  # astype wraps (C cast semantics)


def step_i8(state, tokens, shift: int = 1):
    """One env step, k=1, T=1 (== get_child_states + done_per_game).
    state int8 (B,S,S,S); tokens int8 (B,3S).  Returns (new_state int8 wrapped,
    done uint8 (B,), overflow uint8 (B,))."""
    state = np.asarray(state)
    assert state.dtype == np.int8
    B = state.shape[0]
    new64 = get_child_states(state[:, None], np.asarray(tokens)[:, None, :], shift)[0][:, 0]
    new8, ovf = _narrow_i8(new64)
    return new8, done_per_game(new8).astype(np.uint8).reshape(B), ovf


def step_many_i8(state, tokens, shift: int = 1):
    """K sequential steps (== reference take_actions, datasets.py:144-153, applied per game,
    narrowed to int8 after every step exactly as K calls of step_i8 would).
    tokens int8 (B,K,3S).  Returns (final int8, done_step int32 (B,) = first k whose
    post-state is all zero else -1, overflow uint8 sticky over the K steps)."""
    state = np.asarray(state)
    tokens = np.asarray(tokens)
    B, K = tokens.shape[:2]
    done_step = np.full(B, -1, np.int32)
    ovf = np.zeros(B, np.uint8)
    cur = state.copy()
    for k in range(K):
        cur, d, o = step_i8(cur, tokens[:, k], shift)
        ovf |= o
        done_step = np.where((done_step < 0) & (d != 0), np.int32(k), done_step)
    return cur, done_step, ovf


def expand_i8(state, tokens, shift: int = 1):
    """k children per parent (== get_child_states k>1, T=1) + remove_null_actions per game.
    state int8 (B,S,S,S); tokens int8 (B,k,3S).  Returns (children int8 (B,k,S,S,S),
    done uint8 (B,k), changed uint8 (B,k), overflow uint8 (B,k))."""
    state = np.asarray(state)
    tokens = np.asarray(tokens)
    B, k = tokens.shape[:2]
    S = state.shape[-1]
    kids = get_child_states(state[:, None], tokens, shift)  # list of k (B,1,S,S,S)
    kids64 = np.stack([c[:, 0] for c in kids], axis=1)  # (B,k,S,S,S)
    flat, ovf = _narrow_i8(kids64.reshape(B * k, S, S, S))
    kids8 = flat.reshape(B, k, S, S, S)
    done = done_per_game(flat).astype(np.uint8).reshape(B, k)
    changed = (kids64 != state[:, None].astype(np.int64)).reshape(B, k, -1).any(axis=2).astype(np.uint8)
    return kids8, done, changed, ovf.reshape(B, k)


def gen_from_factors_i8(tokens, shift: int = 1):
    """Deterministic half of the generator (== reference create_synthetic_demo's
    ``target += tensor_action`` loop, utils.py:218-232 / datasets.py:127-141; == uvw_to_demo).
    tokens int8 (B,R,3S) -> (target int8 (B,S,S,S) wrapped, overflow uint8 (B,)).
    The sum is taken in wide arithmetic and narrowed once (sum of int8 wraps == wrap of sum)."""
    tokens = np.asarray(tokens)
    tgt64 = action_to_tensor(tokens, shift).sum(axis=1)
    return _narrow_i8(tgt64)


def reset_matmul_i8(B: int, n: int):
    """Every game <- <n,n,n> (== build_matmul_tensor(1,n,n,n)[0], utils.py:143-161)."""
    t = build_matmul_tensor(1, n, n, n)[0].astype(np.int8)
    return np.broadcast_to(t, (B,) + t.shape).copy()


# ---- Philox-4x32-10 (Salmon et al., SC'11; counter-based, identical on CPU and GPU) ----

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr (...,4) uint32, key (...,2) uint32 -> (...,4) uint32."""
    c = [np.asarray(ctr)[..., i].astype(np.uint64) for i in range(4)]
    k0 = np.asarray(key)[..., 0].astype(np.uint64)
    k1 = np.asarray(key)[..., 1].astype(np.uint64)
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c = [(hi1 ^ c[1] ^ k0) & _MASK, lo1, (hi0 ^ c[3] ^ k1) & _MASK, lo0]
        k0 = (k0 + np.uint64(_W0)) & _MASK
        k1 = (k1 + np.uint64(_W1)) & _MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def categorical_thresholds(probs) -> np.ndarray:
    """uint32 cdf thresholds for a categorical draw from one 32-bit uniform ``d``:
    value index = number of thresholds t with d >= t.  ``probs`` are normalised like
    torch's Categorical (reference utils.py:198, datasets.py:156)."""
    p = np.asarray(probs, dtype=np.float64)
    cdf = np.cumsum(p) / p.sum()
    t = np.floor(cdf[:-1] * 4294967296.0 + 0.5)
    return np.minimum(t, 4294967295.0).astype(np.uint32)


GEN_MAX_ATTEMPTS = 1 << 16
STREAM_FACTORS = 0x00000000
STREAM_BASIS = 0x80000000


def thresholds16(thr):
    """The generator draws 16-bit uniforms: a draw d16 selects values[#{t : d16 * 2^16 >= thr_t}], i.e. it is
    compared with ceil(thr_t / 2^16) in [0, 65536] (the 32-bit cdf thresholds quantised upwards to 16 bits:
    every probability is honoured to within 2^-16)."""
    t = np.asarray(thr, np.uint64)
    return ((t + np.uint64(0xFFFF)) >> np.uint64(16)).astype(np.uint32)


def _draw_vector(seed: int, gid, sub: int, attempt, S: int, thr, values):
    """One attempt at one factor vector for every game in ``gid`` (uint64 array).
    Counter = (gid_lo, gid_hi, sub, attempt<<8 | block); key = (seed_lo, seed_hi).  One Philox block yields EIGHT
    16-bit draws: element e = 8*block + 2*m + half comes from output word m (x,y,z,w = 0..3), low half first."""
    nblk = (S + 7) // 8
    B = gid.shape[0]
    ctr = np.zeros((B, nblk, 4), np.uint32)
    ctr[:, :, 0] = (gid & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
    ctr[:, :, 1] = (gid >> np.uint64(32)).astype(np.uint32)[:, None]
    ctr[:, :, 2] = np.uint32(sub)
    ctr[:, :, 3] = (np.asarray(attempt, np.uint32)[:, None] << np.uint32(8)) | np.arange(nblk, dtype=np.uint32)[None, :]
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    w = philox4x32_10(ctr, np.broadcast_to(key, (B, nblk, 2)))            # (B, nblk, 4) words
    d = np.stack([w & np.uint32(0xFFFF), w >> np.uint32(16)], axis=-1)     # (B, nblk, 4, 2): low half first
    d = d.reshape(B, nblk * 8)[:, :S]
    idx = (d[:, :, None] >= thresholds16(thr)[None, None, :]).sum(axis=2)
    return np.asarray(values, np.int64)[idx]


def gen_factors(B: int, S: int, R: int, thresholds, values, seed: int, game_id_offset: int = 0):
    """Factor vectors of the synthetic demos (reference utils.py:220-231 /
    datasets.py:129-140 with the build's counter-based RNG): for each term, each of
    u,v,w is redrawn until it is not the zero vector (per-vector rejection == the
    reference's joint rejection in distribution).  Returns int64 (B,R,3,S)."""
    values = np.asarray(values, np.int64)
    gid = np.arange(B, dtype=np.uint64) + np.uint64(game_id_offset)
    out = np.zeros((B, R, 3, S), np.int64)
    for r in range(R):
        for x in range(3):
            sub = STREAM_FACTORS | (r * 3 + x)
            attempt = np.zeros(B, np.uint32)
            vec = _draw_vector(seed, gid, sub, attempt, S, thresholds, values)
            bad = ~(vec != 0).any(axis=1)
            while bad.any():
                attempt[bad] += 1
                if int(attempt.max()) >= GEN_MAX_ATTEMPTS:
                    raise RuntimeError("generator: zero vector after GEN_MAX_ATTEMPTS draws")
                redo = _draw_vector(seed, gid[bad], sub, attempt[bad], S, thresholds, values)
                vec[bad] = redo
                bad = ~(vec != 0).any(axis=1)
            out[:, r, x] = vec
    return out


def gen_demos_i8(B, S, R, thresholds, values, shift, seed, game_id_offset=0, basis=None):
    """The generator: tokens int8 (B,R,3S) = cat(u,v,w)+shift and target int8 (B,S,S,S)
    = sum of the R rank-1 terms (reference utils.py:203-233).  With ``basis`` (B,3,S,S) every
    term is emitted in the new basis, (u,v,w) -> (Au,Bv,Cw) (not in the
    reference).  Tokens are narrowed to int8 with wrap (overflow flagged) and the target is the
    sum of the rank-1 terms of the EMITTED tokens.  Returns (tokens, target, overflow)."""
    f = gen_factors(B, S, R, thresholds, values, seed, game_id_offset)
    if basis is not None:
        f = transform_factors(f, basis)
    tok64 = f.reshape(B, R, 3 * S) + int(shift)
    tok_ovf = ((tok64 < -128) | (tok64 > 127)).reshape(B, -1).any(axis=1).astype(np.uint8)
    tokens = tok64.astype(np.int8)
    target, ovf = gen_from_factors_i8(tokens, shift)
    return tokens, target, ovf | tok_ovf


# ---- change of basis (not in the reference; paper-level spec; PARITY UNPINNED) ----


def sample_basis(B, S, thresholds, values, seed, game_id_offset=0):
    """Three unimodular matrices per game, P = L @ U with L unit(+-1)-diagonal lower and
    U unit(+-1)-diagonal upper triangular, off-diagonal entries categorical over ``values``.
    One 32-bit draw per cell (a,b) of each matrix: a>b -> L[a,b]; a<b -> U[a,b];
    a==b -> bit0 = sign of L[a,a], bit1 = sign of U[a,a].
    Counter = (gid_lo, gid_hi, STREAM_BASIS | mode, block) with block = (a*S+b)//4.
    Values must satisfy S * v^2 <= 127 (P is emitted as int8; the library refuses larger ones).
    Returns (P int64 (B,3,S,S), L, U)."""
    values = np.asarray(values, np.int64)
    if (S * values * values > 127).any():
        raise ValueError("sample_basis: need S * v^2 <= 127 for every value (P = L @ U must fit int8)")
    thr = np.asarray(thresholds, np.uint32)
    gid = np.arange(B, dtype=np.uint64) + np.uint64(game_id_offset)
    nblk = (S * S + 3) // 4
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    L = np.zeros((B, 3, S, S), np.int64)
    U = np.zeros((B, 3, S, S), np.int64)
    a_idx, b_idx = np.divmod(np.arange(S * S), S)
    for x in range(3):
        ctr = np.zeros((B, nblk, 4), np.uint32)
        ctr[:, :, 0] = (gid & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
        ctr[:, :, 1] = (gid >> np.uint64(32)).astype(np.uint32)[:, None]
        ctr[:, :, 2] = np.uint32(STREAM_BASIS | x)
        ctr[:, :, 3] = np.arange(nblk, dtype=np.uint32)[None, :]
        d = philox4x32_10(ctr, np.broadcast_to(key, (B, nblk, 2))).reshape(B, nblk * 4)[:, : S * S]
        val = values[(d[:, :, None] >= thr[None, None, :]).sum(axis=2)]  # (B,S*S)
        lower, upper, diag = a_idx > b_idx, a_idx < b_idx, a_idx == b_idx
        Lx = np.zeros((B, S * S), np.int64)
        Ux = np.zeros((B, S * S), np.int64)
        Lx[:, lower] = val[:, lower]
        Ux[:, upper] = val[:, upper]
        Lx[:, diag] = 1 - 2 * (d[:, diag] & 1).astype(np.int64)
        Ux[:, diag] = 1 - 2 * ((d[:, diag] >> 1) & 1).astype(np.int64)
        L[:, x] = Lx.reshape(B, S, S)
        U[:, x] = Ux.reshape(B, S, S)
    return L @ U, L, U


def change_basis_i8(state, basis):
    """T'[a,b,c] = sum_ijk A[a,i] B[b,j] C[c,k] T[i,j,k]; basis int (B,3,S,S) = (A,B,C).
    Returns (int8 wrapped, overflow uint8 (B,))."""
    T = np.asarray(state).astype(np.int64)
    M = np.asarray(basis).astype(np.int64)
    out = np.einsum("nai,nbj,nck,nijk->nabc", M[:, 0], M[:, 1], M[:, 2], T, optimize=True)
    return _narrow_i8(out)


def transform_factors(factors, basis):
    """(u,v,w) -> (A u, B v, C w) per term; factors int (B,R,3,S), basis (B,3,S,S)."""
    f = np.asarray(factors).astype(np.int64)
    M = np.asarray(basis).astype(np.int64)
    return np.einsum("nxai,nrxi->nrxa", M, f)


def unimodular_inverse(L, U):
    """Exact integer inverse of P = L @ U for unit(+-1)-diagonal triangular L, U
    (forward/back substitution; no division other than by +-1)."""
    L = np.asarray(L).astype(np.int64)
    U = np.asarray(U).astype(np.int64)
    S = L.shape[-1]

    def inv_lower(M):
        X = np.zeros_like(M)
        for c in range(S):
            for r in range(c, S):
                acc = (1 if r == c else 0) - (M[..., r, c:r] * X[..., c:r, c]).sum(axis=-1)
                X[..., r, c] = acc * M[..., r, r]  # 1/(+-1) == +-1
        return X

    Linv = inv_lower(L)
    Uinv = np.swapaxes(inv_lower(np.swapaxes(U, -1, -2)), -1, -2)
    return Uinv @ Linv


# ---- next rows ----------------------------------------------------------


def model_input(frames_newest_first, t_step, dtype=np.float32):
    """N1: what the reference feeds the model after a step -- the (B,T,S,S,S) float state whose
    frame 0 is the new head and frames 1.. the previous frames (act.py:271-274), and
    get_scalars (utils.py:22-37): a (B,1) float tensor filled with t_step."""
    f = np.asarray(frames_newest_first)
    return f.astype(dtype), np.full((f.shape[0], 1), float(t_step), np.float32)


def _fmix64(k):
    k = np.asarray(k, np.uint64)
    k = k ^ (k >> np.uint64(33))
    k = k * np.uint64(0xFF51AFD7ED558CCD)
    k = k ^ (k >> np.uint64(33))
    k = k * np.uint64(0xC4CEB9FE1A85EC53)
    k = k ^ (k >> np.uint64(33))
    return k


def state_hash(state) -> np.ndarray:
    """N2: the 64-bit key of include/tensor_game.h::tg_hash_u64 (replaces state_to_str,
    utils.py:164-169).  Returns uint64 (B,)."""
    st = np.ascontiguousarray(np.asarray(state).astype(np.int8))
    B = st.shape[0]
    N = st[0].size
    nword = (N + 7) // 8
    buf = np.zeros((B, nword * 8), np.uint8)
    buf[:, :N] = st.reshape(B, N).view(np.uint8)
    words = buf.view("<u8").reshape(B, nword)
    with np.errstate(over="ignore"):
        salt = (np.arange(1, nword + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))[None, :]
        h = _fmix64(words + salt).sum(axis=1, dtype=np.uint64)
        return _fmix64(h ^ (np.uint64(N) * np.uint64(0xC2B2AE3D27D4EB4F)))


def seen_u64(keys, table: set, mask=None, insert=False) -> np.ndarray:
    """N2, second half (include/tensor_game.h::tg_seen_u64): fresh[i] = mask[i] and keys[i] not in `table`, decided
    against the table as it was BEFORE the call; with `insert` every masked key is in the table afterwards.  `table`
    is a Python set of ints -- exactly the role of `new_mc_tree`'s keys in extend_tree (act.py:188-195: `c not in
    new_mc_tree`; act.py:209-211: the expanded state enters the tree), on 64-bit keys instead of state_to_str strings."""
    k = np.asarray(keys, np.uint64)
    m = np.ones(k.shape, bool) if mask is None else np.asarray(mask).astype(bool)
    flat = [int(x) for x in k.reshape(-1)]
    fresh = np.array([mm and (x not in table) for x, mm in zip(flat, m.reshape(-1))], np.uint8).reshape(k.shape)
    if insert:
        table.update(x for x, mm in zip(flat, m.reshape(-1)) if mm)
    return fresh


def tree_filter(parent, actions, table: set, shift=1):
    """The candidate filter of extend_tree (act.py:183-195) for ONE expanded state: children = parent - tensor(action)
    (get_child_states, act.py:266-275), drop the children equal to the parent (remove_null_actions, utils.py:191-194),
    drop those whose key is already a key of the tree.  parent int8 (S,S,S), actions (k,3S) tokens.
    Returns (kept uint8 (k,), child keys uint64 (k,), changed uint8 (k,))."""
    parent = np.asarray(parent).astype(np.int64)
    actions = np.asarray(actions)
    kids = parent[None] - action_to_tensor(actions, shift)
    changed = (kids != parent[None]).reshape(len(actions), -1).any(axis=1).astype(np.uint8)
    keys = state_hash(kids.astype(np.int8))
    return seen_u64(keys, table, mask=changed), keys, changed


def _rank_bareiss(m):
    """Exact rational rank of an integer matrix (list of lists of Python ints): Bareiss
    fraction-free elimination -- every division is exact, entries stay minors of the input."""
    n_rows, n_cols = len(m), len(m[0])
    m = [row[:] for row in m]
    rank, prev = 0, 1
    for c in range(n_cols):
        piv = next((r for r in range(rank, n_rows) if m[r][c] != 0), None)
        if piv is None:
            continue
        m[rank], m[piv] = m[piv], m[rank]
        p = m[rank][c]
        for r in range(rank + 1, n_rows):
            f = m[r][c]
            m[r] = [(p * x - f * y) // prev for x, y in zip(m[r], m[rank])]
        prev = p
        rank += 1
        if rank == n_rows:
            break
    return rank


def slice_rank_exact(state) -> np.ndarray:
    """N3: sum over slices state[b][i] of the EXACT rational rank of the S x S integer matrix.
    The reference's get_rank (utils.py:134-140) is the float-SVD version of the same quantity;
    they agree on the golden states (tests/golden/next_rows.npz)."""
    st = np.asarray(state).astype(np.int64)
    B, S = st.shape[0], st.shape[1]
    out = np.zeros(B, np.int32)
    for b in range(B):
        out[b] = sum(_rank_bareiss([[int(x) for x in row] for row in st[b, i]]) for i in range(S))
    return out
