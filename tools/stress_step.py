#!/usr/bin/env python3
"""Randomised differential soak of the single-step kernels (S=16 / S=25 compaction queues and their dense-factor
paths, the whole-line variant, S=4, the staged kernels) and of the streamed stepper against the numpy oracle:
token densities swept so that the candidate count crosses the queue capacity (64), wide factors, large shifts,
states at the int8 edge, null actions, in place / out of place, padded / packed layouts; and of tg_expand_i8 on
the same material (children, done, changed, overflow).
    python tools/stress_step.py [cases]          (TG_LIB_VARIANT=ab TG_S16_LINES=1 ... for the forced variants)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402
from oracle import tensor_game as O  # noqa: E402

DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(2024)
bad = 0


def padded(st):
    t = ops.alloc_states(st.shape[0], st.shape[1], DEV)
    t.copy_(torch.from_numpy(st))
    return t


for c in range(cases):
    S = int(rng.choice([16, 25, 16, 25, 4, 9]))
    B = int(rng.integers(1, 70))
    shift = int(rng.choice([1, 1, 1, 2, 0, -1, 100, 127, -127, 300]))
    dens = float(rng.choice([0.05, 0.15, 0.25, 0.3, 0.4, 0.5, 0.7, 1.0]))  # P(factor != 0): 0.25-0.5 straddles 64 candidates at S=16
    ac = np.where(rng.random((B, 3 * S)) < dens, rng.choice([-1, 1], size=(B, 3 * S)), 0).astype(np.int64)
    kind = c % 7
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    if kind == 1:
        ac[rng.integers(B)] = rng.integers(-5, 6, size=3 * S)                      # moderate factors
    elif kind == 2:
        ac[::2] = rng.integers(-128 - shift, 128 - shift, size=ac[::2].shape)        # anything a token can hold
    elif kind == 3:
        st = rng.choice([-128, -127, 126, 127, 0, 1], size=st.shape).astype(np.int8)  # int8 edge
    elif kind == 4:
        ac[:, rng.integers(3) * S:][:, :S] = 0                                        # a zero vector: null actions
        st[::3] = 0
    elif kind == 5:
        st[::2] = O.gen_from_factors_i8(np.clip(ac[::2, None, :] + 1, -128, 127).astype(np.int8), 1)[0]  # finishes
    tok = np.clip(ac + shift, -128, 127).astype(np.int8)
    want, wdone, wovf = O.step_i8(st, tok, shift=shift)
    ok = True
    for layout in ("padded", "packed"):
        for inplace in (False, True):
            t = padded(st) if layout == "padded" else torch.from_numpy(st).to(DEV)
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            out, done = ops.step(t, torch.from_numpy(tok).to(DEV), out=t if inplace else None, overflow=ovf, shift=shift)
            ok = ok and np.array_equal(out.cpu().numpy(), want) and np.array_equal(done.cpu().numpy(), wdone)
            ok = ok and np.array_equal(ovf.cpu().numpy(), wovf)
            ok = ok and (inplace or np.array_equal(t.cpu().numpy(), st))
    # the tracked step (nnz carried; S = 16 / 25: the sparse kernels, forced for every batch under TG_TRACKED_SPARSE) on the
    # same material, two steps of one rollout
    for layout in ("padded", "packed"):
        t = padded(st) if layout == "padded" else torch.from_numpy(st).to(DEV)
        _, nnz = ops.done(t, want_nnz=True)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        cur, wo2 = st, np.zeros(B, np.uint8)
        for k in range(2):
            tk = np.roll(tok, k, axis=0)
            cur, d2, o2 = O.step_i8(cur, tk, shift=shift)
            wo2 |= o2
            _, done = ops.step_tracked(t, torch.from_numpy(tk).to(DEV), nnz, overflow=ovf, shift=shift)
            ok = ok and np.array_equal(t.cpu().numpy(), cur) and np.array_equal(done.cpu().numpy(), d2)
            ok = ok and np.array_equal(nnz.cpu().numpy(), np.count_nonzero(cur.reshape(B, -1), axis=1))
        ok = ok and np.array_equal(ovf.cpu().numpy(), wo2)
    if S in (4, 16, 25) and abs(shift) <= 127:  # the streamed stepper (|shift| <= 127 only): K steps of the same kind
        K = int(rng.integers(1, 21))                                                    # (blocks of up to 8 steps)
        toks = np.stack([np.roll(tok, k, axis=0) for k in range(K)])
        cur, wd, wo = st.copy(), np.zeros((K, B), np.uint8), np.zeros(B, np.uint8)
        for k in range(K):
            cur, d, o = O.step_i8(cur, toks[k], shift=shift)
            wd[k] = d
            wo |= o
        t = padded(st)
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        gated = bool(c & 1)                                                             # ready words (all set) and progress words
        ready = torch.ones(K, dtype=torch.int32, device=DEV) if gated else None
        prog = torch.zeros(ops.step_stream_layout(B, S, DEV)[0], dtype=torch.int32, device=DEV) if gated else None
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        _, done = ops.step_stream(t, torch.from_numpy(toks).to(DEV), overflow=ovf, ready=ready, progress=prog, status=status, shift=shift)
        torch.cuda.synchronize()
        ok = ok and np.array_equal(t.cpu().numpy(), cur) and np.array_equal(done.cpu().numpy(), wd)
        ok = ok and np.array_equal(ovf.cpu().numpy(), wo) and int(status[0]) == 0 and (prog is None or bool((prog == K).all()))
    # tg_expand_i8 on the same material: k children per parent (the game's own action and its neighbours')
    k = int(rng.integers(1, 9))
    ak = np.stack([np.roll(tok, j, axis=0) for j in range(k)], axis=1)
    wk, wd, wc, wo = O.expand_i8(st, ak, shift)
    for layout in ("padded", "packed"):
        t = padded(st) if layout == "padded" else torch.from_numpy(st).to(DEV)
        ovk = torch.zeros((B, k), dtype=torch.uint8, device=DEV)
        kids, d, ch = ops.expand(t, torch.from_numpy(ak).to(DEV), overflow=ovk, shift=shift)
        ok = ok and np.array_equal(kids.cpu().numpy(), wk) and np.array_equal(d.cpu().numpy(), wd)
        ok = ok and np.array_equal(ch.cpu().numpy(), wc) and np.array_equal(ovk.cpu().numpy(), wo)
        ok = ok and np.array_equal(t.cpu().numpy(), st)
    # round 4: keys formed inside the expansion (S = 4, 16) and step + model input in one kernel (S = 4, 16), same material
    if S in (4, 16, 25) and abs(shift) <= 127:
        kids, d, ch, keys = ops.expand(padded(st), torch.from_numpy(ak).to(DEV), shift=shift, want_keys=True)
        ok = ok and np.array_equal(kids.cpu().numpy(), wk)
        ok = ok and np.array_equal(keys.cpu().numpy().view(np.uint64), O.state_hash(wk.reshape(B * k, S, S, S)).reshape(B, k))
    if S in (4, 16) and abs(shift) <= 127:
        T = int(rng.integers(1, 5))
        frames = rng.integers(-3, 4, size=(B, T, S, S, S)).astype(np.int8)
        head = int(rng.integers(T))
        frames[:, head] = st
        ring = ops.alloc_ring(B, S, T, DEV)
        ring.copy_(torch.from_numpy(frames).to(DEV))
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        dt = [torch.float32, torch.float16, torch.bfloat16][c % 3]
        x, sc, dn, nxt = ops.step_emit(ring, head, torch.from_numpy(tok).to(DEV), 2.0, dtype=dt, overflow=ovf, shift=shift)
        wantr = frames.copy()
        wantr[:, (head + 1) % T] = want
        order = [((head + 1) % T - f) % T for f in range(T)]
        ok = ok and np.array_equal(ring.cpu().numpy(), wantr) and np.array_equal(x.float().cpu().numpy(), wantr[:, order].astype(np.float32))
        ok = ok and np.array_equal(dn.cpu().numpy(), wdone) and np.array_equal(ovf.cpu().numpy(), wovf)
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=c, S=S, B=B, shift=shift, dens=dens, kind=kind), flush=True)
    if c % 50 == 0:
        print("case", c, "bad", bad, flush=True)
print("done", cases, "bad", bad)
sys.exit(1 if bad else 0)
