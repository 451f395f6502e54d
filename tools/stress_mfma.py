#!/usr/bin/env python3
"""Randomised differential soak of the matrix-core paths against the numpy oracle: step_many with long action
lists (all verdict kinds: replays that end at K-1, early ends, never, overflow in the middle, wide factors,
{-2..2} vocabulary), gen_from_factors with R up to 256, gen_demos in a random basis.
    python tools/stress_mfma.py [cases]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mat_mul_amd import ops  # noqa: E402
from oracle import tensor_game as O  # noqa: E402

DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(12345)
thr = O.categorical_thresholds((0.15, 0.7, 0.15))
bad = 0
for c in range(cases):
    S = int(rng.choice([9, 16, 25]))
    B = int(rng.integers(1, 9))
    K = int(rng.integers(1, 128)) if c % 3 else int(rng.integers(100, 257))
    kind = c % 6
    tok, tgt, _ = O.gen_demos_i8(B, S, K, thr, (-1, 0, 1), 1, seed=c)
    st, ac = tgt.copy(), tok.copy()
    if kind == 1:      # never ends: random start
        st = rng.integers(-3, 4, size=st.shape).astype(np.int8)
    elif kind == 2 and K >= 3:    # tail cancels: ends early
        ac[:, K - 2] = tok[:, 0]
        ac[:, K - 1] = tok[:, 0]
        ac[:, K - 1, :S] = 2 - tok[:, 0, :S]
        st = O.gen_from_factors_i8(ac[:, :K - 2], 1)[0]
    elif kind == 3:    # {-2..2} vocabulary
        ac = rng.integers(-1, 4, size=ac.shape).astype(np.int8)
    elif kind == 4:    # near the int8 edge: overflow likely
        st = np.clip(st.astype(int) + rng.integers(-125, 126), -128, 127).astype(np.int8)
    elif kind == 5:    # a few wide factors
        ac[rng.integers(B), rng.integers(K), rng.integers(3 * S)] = np.int8(rng.integers(-128, 128))
    want, wds, wovf = O.step_many_i8(st, ac)
    t = ops.alloc_states(B, S, DEV)
    t.copy_(torch.from_numpy(st))
    ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
    out, ds = ops.step_many(t, torch.from_numpy(ac).to(DEV), overflow=ovf)
    ok = (np.array_equal(out.cpu().numpy(), want) and np.array_equal(ds.cpu().numpy(), wds)
          and np.array_equal(ovf.cpu().numpy(), wovf))
    w2, o2 = O.gen_from_factors_i8(ac, 1)
    ovf.zero_()
    g = ops.gen_from_factors(torch.from_numpy(ac).to(DEV), S, overflow=ovf)
    ok = ok and np.array_equal(g.cpu().numpy(), w2) and np.array_equal(ovf.cpu().numpy(), o2)
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=c, S=S, B=B, K=K, kind=kind), flush=True)
    if c % 50 == 0:
        print("case", c, "bad", bad, flush=True)
print("done", cases, "bad", bad)
sys.exit(1 if bad else 0)
