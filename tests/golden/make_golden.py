#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run once, in the build container (where /root/reference exists), from any cwd:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports kurtosis/mat_mul's own ``utils``, ``datasets`` and ``act`` modules from
``/root/reference`` (read-only; nothing is copied), calls the hot-path functions on
seeded inputs and stores INPUTS and the reference's OUTPUTS as small ``.npz`` files.
The ``.npz`` files are committed; the reference is not needed (and does not exist) on
the GPU box.  Every array is integer-valued (the reference computes in float32/int64 on
small integers, which is exact) and is stored as int8/int16/int64.

Reference functions exercised (file:line in /root/reference):
  utils.py:40-53   uvw_to_demo            datasets.py:423-465 get_strassen_factors/tensor
  utils.py:56-96   action_to_uvw/uvw_to_tensor/action_to_tensor
  utils.py:99-111  get_head_state         utils.py:181-188   tensor_factorized
  utils.py:143-161 build_matmul_tensor    utils.py:191-194   remove_null_actions
  utils.py:197-233 factor_sample/create_synthetic_demo
  act.py:266-275   get_child_states       datasets.py:20-158 SyntheticDemoDataset
  datasets.py:362-420 StrassenDemoDataset
"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def main():
    sys.dont_write_bytecode = True
    os.chdir(tempfile.mkdtemp(prefix="golden_"))
    sys.path.insert(0, REF)
    import torch

    import act  # noqa: E402  (reference)
    import datasets  # noqa: E402  (reference)
    import utils  # noqa: E402  (reference)

    torch.manual_seed(1234)
    rng = np.random.default_rng(20261004)

    def i8(t):
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        assert np.all(a == np.round(a)) and a.min() >= -128 and a.max() <= 127
        return a.astype(np.int8)

    # ---------------------------------------------------------------- 1. Strassen (BASELINE config 1)
    uu, vv, ww = datasets.get_strassen_factors("cpu")
    tensor, tokens = datasets.get_strassen_tensor("cpu")  # fp32 (4,4,4), int64 (7,12) shift=1
    state = tensor.view(1, 1, 4, 4, 4)
    replay = [i8(state[0, 0])]
    done = [bool(utils.tensor_factorized(utils.get_head_state(state)))]
    for k in range(7):
        state = act.get_child_states(state, tokens[k].view(1, 1, 12))[0]
        replay.append(i8(state[0, 0]))
        done.append(bool(utils.tensor_factorized(utils.get_head_state(state))))
    ds = datasets.StrassenDemoDataset()
    np.savez_compressed(
        OUT / "strassen.npz",
        uu=i8(uu), vv=i8(vv), ww=i8(ww), tensor=i8(tensor), tokens=i8(tokens),
        replay=np.stack(replay), done=np.array(done, np.uint8),
        matmul_2=i8(utils.build_matmul_tensor(1, 2, 2, 2)[0]),
        ds_states=np.stack([i8(s[0]) for s in ds.state_tensor]),       # (448,4,4,4)
        ds_actions=np.stack([i8(a) for a in ds.target_action]),        # (448,12) shift=2
        ds_rewards=np.array([int(r.item()) for r in ds.reward], np.int16),
    )

    # ---------------------------------------------------------------- 2. build_matmul_tensor n=2..5
    mm = {f"n{n}_t{T}": i8(utils.build_matmul_tensor(T, n, n, n)) for n in (2, 3, 4, 5) for T in (1, 2)}
    np.savez_compressed(OUT / "matmul_tensors.npz", **mm)

    # ---------------------------------------------------------------- 3. get_child_states / tensor_factorized
    step = {}
    cases = [(4, 1, 1, 1), (4, 3, 2, 5), (4, 257, 1, 1), (4, 64, 3, 8), (9, 3, 2, 5), (9, 65, 1, 1),
             (16, 1, 1, 1), (16, 17, 2, 3), (25, 1, 1, 1), (25, 4, 2, 2), (5, 7, 1, 3), (8, 9, 1, 2)]
    for (S, B, T, k) in cases:
        tag = f"S{S}_B{B}_T{T}_k{k}"
        st = torch.from_numpy(rng.integers(-2, 3, size=(B, T, S, S, S)).astype(np.float32))
        ac = torch.from_numpy(rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, k, 3 * S)).astype(np.int64))
        # terminal games: head == action tensor of candidate 0; null action: candidate k-1 has u == 0
        for b in range(0, B, 3):
            while (utils.action_to_tensor(ac[b, 0]) == 0).all():
                ac[b, 0] = torch.from_numpy(rng.integers(0, 3, size=3 * S))
            st[b, 0] = utils.action_to_tensor(ac[b, 0]).float()
        for b in range(1, B, 4):
            ac[b, k - 1, :S] = 1  # u = 0 (shift 1) -> null action
        kids = act.get_child_states(st, ac)
        assert len(kids) == k
        step[tag + "_state"] = i8(st)
        step[tag + "_actions"] = i8(ac)
        step[tag + "_children"] = np.stack([i8(c) for c in kids], axis=1)  # (B,k,T,S,S,S)
        step[tag + "_done"] = np.array(
            [[bool(utils.tensor_factorized(utils.get_head_state(c[b:b + 1]))) for c in kids] for b in range(B)],
            np.uint8)
        # remove_null_actions is defined over the whole batch (utils.py:193); record it per game too
        step[tag + "_nonnull_batch"] = np.array(utils.remove_null_actions(st, kids), np.int64)
        step[tag + "_changed"] = np.array(
            [[int(i in utils.remove_null_actions(st[b:b + 1], [c[b:b + 1] for c in kids])) for i in range(k)]
             for b in range(B)], np.uint8)
        # the verbatim (game-0-only) tensor_factorized on the full batched state, SURVEY section 0
        step[tag + "_tf_verbatim"] = np.array([bool(utils.tensor_factorized(c)) for c in kids], np.uint8)
    np.savez_compressed(OUT / "step_cases.npz", **step)

    # ---------------------------------------------------------------- 4. action_to_tensor single vs batched, shifts
    a2t = {}
    for S in (4, 9, 16, 25):
        ac = torch.from_numpy(rng.integers(0, 3, size=(6, 3 * S)).astype(np.int64))
        a2t[f"S{S}_actions"] = i8(ac)
        a2t[f"S{S}_batched"] = i8(utils.action_to_tensor(ac))
        a2t[f"S{S}_single"] = np.stack([i8(utils.action_to_tensor(a)) for a in ac])
        u, v, w = utils.action_to_uvw(ac, shift=2)
        a2t[f"S{S}_shift2"] = i8(utils.uvw_to_tensor((u, v, w)))
    # wider factor range (tokens are never range-checked, utils.py:64-66)
    ac = torch.from_numpy(rng.integers(-2, 6, size=(5, 12)).astype(np.int64))
    a2t["wide_actions"] = i8(ac)
    a2t["wide_tensor"] = utils.action_to_tensor(ac).numpy().astype(np.int16)
    np.savez_compressed(OUT / "action_to_tensor.npz", **a2t)

    # ---------------------------------------------------------------- 5. synthetic demos + _take_actions + __getitem__
    syn = {}
    values = torch.tensor((-1, 0, 1))
    for (S, R, n) in [(4, 7, 6), (9, 12, 3), (16, 20, 2), (25, 30, 1)]:
        probs = torch.tensor((0.15, 0.7, 0.15))
        for d in range(n):
            seq, tgt = utils.create_synthetic_demo(values, probs, R, S, 1)
            syn[f"fn_S{S}_R{R}_{d}_tokens"] = np.stack([i8(a) for a in seq])
            syn[f"fn_S{S}_R{R}_{d}_target"] = i8(tgt)
    for (S, R, n, T) in [(4, 7, 4, 3), (9, 10, 2, 4), (16, 12, 1, 2)]:
        save_dir = Path(tempfile.mkdtemp(prefix="synth_"))
        dset = datasets.SyntheticDemoDataset(R, n, T, S, "cpu", save_dir=save_dir)
        for d in range(n):
            seq = torch.load(save_dir / f"action_seq_{d}.pt")
            tgt = torch.load(save_dir / f"target_tensor_{d}.pt")
            syn[f"ds_S{S}_R{R}_T{T}_{d}_tokens"] = np.stack([i8(a) for a in seq])
            syn[f"ds_S{S}_R{R}_T{T}_{d}_target"] = i8(tgt)
            # _take_actions on every suffix (datasets.py:90-92)
            syn[f"ds_S{S}_R{R}_T{T}_{d}_suffix_states"] = np.stack(
                [i8(dset._take_actions(seq[i + 1:], tgt)) for i in range(R)])
            for ia in range(R):
                frames, scalar, action, reward = dset[d * R + ia]
                syn[f"ds_S{S}_R{R}_T{T}_{d}_item{ia}_frames"] = i8(frames)
                syn[f"ds_S{S}_R{R}_T{T}_{d}_item{ia}_meta"] = np.array(
                    [scalar.item(), reward.item()], np.float32)
                syn[f"ds_S{S}_R{R}_T{T}_{d}_item{ia}_action"] = i8(action)
    np.savez_compressed(OUT / "synthetic_demos.npz", **syn)

    # ---------------------------------------------------------------- 6. sampler statistics (distribution target)
    stats = {}
    for (S, probs) in [(4, (0.15, 0.7, 0.15)), (9, (0.15, 0.7, 0.15)), (4, (0.1, 0.8, 0.1))]:
        p = torch.tensor(probs)
        n_terms, counts, attempts = 4000, np.zeros(3, np.int64), 0
        for _ in range(n_terms):
            while True:
                attempts += 1
                vecs = [utils.factor_sample(values, p, S) for _ in range(3)]
                if not (utils.uvw_to_tensor(tuple(vecs)) == 0).all():
                    break
            for vec in vecs:
                for val in (-1, 0, 1):
                    counts[val + 1] += int((vec == val).sum())
        key = f"S{S}_p{int(probs[1] * 100)}"
        stats[key + "_value_counts"] = counts
        stats[key + "_terms_attempts"] = np.array([n_terms, attempts], np.int64)
    np.savez_compressed(OUT / "sampler_stats.npz", **stats)

    # ---------------------------------------------------------------- 7. get_rank (terminal reward), multi-step histories
    nxt = {}
    for (S, B) in [(4, 40), (9, 12), (16, 4), (25, 2)]:
        st = torch.from_numpy(rng.integers(-1, 2, size=(B, 1, S, S, S)).astype(np.float32))
        st[::3] *= (torch.from_numpy(rng.random((len(st[::3]), 1, S, S, S))) < 0.15).float()   # sparse states
        for b in range(1, B, 4):                                                        # low-rank states
            acs = torch.from_numpy(rng.integers(0, 3, size=(3, 3 * S)).astype(np.int64))
            st[b, 0] = utils.action_to_tensor(acs).sum(0).float()
        nxt[f"rank_S{S}_state"] = i8(st[:, 0])
        nxt[f"rank_S{S}_rank"] = np.array([utils.get_rank(st[b:b + 1]) for b in range(B)], np.int32)
    # a 5-step rollout with T=3 history through get_child_states (k=1), recording every state
    for (S, B, T) in [(4, 6, 3), (9, 3, 2)]:
        st = torch.zeros((B, T, S, S, S))
        st[:, 0] = torch.from_numpy(rng.integers(-1, 2, size=(B, S, S, S)).astype(np.float32))
        seq, acts = [i8(st)], []
        for step in range(5):
            ac = torch.from_numpy(rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 1, 3 * S)).astype(np.int64))
            st = act.get_child_states(st, ac)[0]
            seq.append(i8(st))
            acts.append(i8(ac[:, 0]))
            sc = utils.get_scalars(st, step + 1)
            assert sc.shape == (B, 1) and float(sc[0, 0]) == step + 1
        nxt[f"hist_S{S}_T{T}_states"] = np.stack(seq)       # (6,B,T,S,S,S)
        nxt[f"hist_S{S}_T{T}_actions"] = np.stack(acts)     # (5,B,3S)
    np.savez_compressed(OUT / "next_rows.npz", **nxt)

    total = sum(f.stat().st_size for f in OUT.glob("*.npz"))
    print(f"wrote {len(list(OUT.glob('*.npz')))} fixtures, {total / 1024:.0f} KiB total")


if __name__ == "__main__":
    main()
