#!/usr/bin/env python3
"""bench.py -- env steps/s of the tensor-game step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is ONE launch of the hot-path kernel (tg_step_i8, in place) over the whole batch of
games resident in HBM.  Workload at every N: BASELINE config 2 per GPU -- S=4 int8, 65 536
independent games per GPU (weak scaling: games are sharded by contiguous global id range, no
collective on the data path).  The K timed steps are chained (each step consumes the state the
previous one wrote) and cycle through a 2R-action schedule that returns every game to its start
state, so the timed region checks itself.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- algorithmic bytes per launch / average launch time (HIP events on the launch
                  stream) against the 8 TB/s HBM peak, for the kernel the timed region runs;
  cpu_baseline -- the oracle's reference-dtype torch-CPU port (oracle/ref_dtype_torch.py) timed
                  on this box's host cores on a bounded sample of the same workload;
  also         -- the same measurement on the other single-GPU BASELINE configs and on a
                  batch large enough to stream from HBM (informational, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def bytes_step(S: int) -> int:
    """read state + read action + write state + write done."""
    return 2 * S ** 3 + 3 * S + 1


def make_schedule(B, S, R, dev, seed, gid0):
    """Start state + 2R token tensors: the demo's own R actions, then the same actions with u
    negated (which add the terms back).  After R steps every game is zero; after 2R it is back."""
    from mat_mul_amd import ops

    actions, target = ops.gen_demos(B, S, R, dev, seed=seed, game_id_offset=gid0)
    sched = []
    for k in range(R):
        sched.append(actions[:, k].contiguous())
    for k in range(R):
        a = actions[:, k].clone()
        a[:, :S] = 2 - a[:, :S]  # token = u + 1  ->  -u + 1
        sched.append(a.contiguous())
    return target, sched


def time_steps(B, S, K, W, dev, mode, seed=0, gid0=0, R=None, sync=None):
    """Returns dict(wall_s, event_ms, ok).  EXACTLY K timed launches after W warm-up launches."""
    from mat_mul_amd import ops

    R = R or (7 if S == 4 else 8)
    target, sched = make_schedule(B, S, R, dev, seed, gid0)
    state = ops.alloc_states(B, S, dev)
    state.copy_(target)
    done = torch.zeros(B, dtype=torch.uint8, device=dev)
    ovf = torch.zeros(B, dtype=torch.uint8, device=dev)
    launch = ops.prepare_step(state, sched, done, ovf, shift=1)
    L = len(sched)
    pos = 0
    for _ in range(W):
        launch(pos % L)
        pos += 1
    torch.cuda.synchronize(dev)
    start_pos = pos

    plan = []  # (graph, replays, kernel nodes) in timed order
    warm_replay = 0
    if mode == "graph":
        # Graphs hold whole 2R cycles wherever possible: such a graph is the same for every chunk (one
        # instantiation, replayed).  Every graph is replayed once UNTIMED before the timed region, because the
        # first replay of a hipGraph also uploads it (one-off, ~0.1 us per node), which is not part of a step.
        CH = max(L, (2048 // L) * L)  # kernel nodes per graph
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))

        def capture(n):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for j in range(n):
                    launch((start_pos + j) % L)  # every full chunk starts at the same phase
            return g

        nfull, rem = divmod(K, CH)
        if nfull:
            plan.append((capture(CH), nfull, CH))
        if rem:
            plan.append((capture(rem), 1, rem))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)  # capture does not execute: the state is still at start_pos
        if not os.environ.get("TG_BENCH_NO_WARM_REPLAY"):
            # every graph once, in timed order (K steps), then eager steps up to the next multiple of 2R: the
            # schedule is cyclic, so the state is back at start_pos's point of it, for any K
            for g, _, n in plan:
                g.replay()
                warm_replay += n
            for j in range((-rem) % L):  # the full-chunk graph is whole cycles; only the remainder leaves a phase
                launch((start_pos + rem + j) % L)
                warm_replay += 1
            torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if sync:
        sync()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record()
    if mode == "graph":
        for g, reps, _ in plan:
            for _ in range(reps):
                g.replay()
    else:
        for k in range(K):
            launch((start_pos + k) % L)
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0  # this rank's K steps; the caller takes the MAX over ranks
    if sync:
        sync()
    pos = start_pos + K
    # self-check of the timed region: finish the current 2R cycle and compare with the start state
    while pos % L:
        launch(pos % L)
        pos += 1
    torch.cuda.synchronize(dev)
    ok = bool(torch.equal(state, target)) and not bool(ovf.any())
    return {"wall_s": wall, "event_ms": ev0.elapsed_time(ev1), "ok": ok, "warm_replay": warm_replay}


def measured_traffic(B, S):
    """HBM bytes per launch of this workload from the committed rocprofv3 PMC passes
    (profiles/traffic_r*.json, written by profiles/summarize.py), newest round first; else None."""
    for f in sorted((ROOT / "profiles").glob("traffic_r*.json"), reverse=True):
        try:
            e = json.loads(f.read_text()).get(f"S{S}_B{B}")
        except (OSError, ValueError):
            continue
        if e and "hbm_bytes_per_launch" in e:
            return e["hbm_bytes_per_launch"], f.name
    return None, None


def roofline(B, S, K, event_ms):
    per_launch_s = event_ms * 1e-3 / K
    achieved = B * bytes_step(S) / per_launch_s / 1e9
    traffic, src = measured_traffic(B, S)
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
            "kernel": {4: "tg::s4_kernel<STEP>", 16: "tg::s16_step_kernel<STEP>", 9: "tg::packed_kernel<9,16,STEP>",
                       25: "tg::packed_kernel<25,256,STEP>"}.get(S, "tg::slow_kernel<STEP>"),
            "bytes_per_launch": B * bytes_step(S), "avg_launch_us": round(per_launch_s * 1e6, 3),
            "note": ("achieved/frac price the ALGORITHMIC bytes 2S^3+3S+1 per step; in-place steps of the S>=9 kernels "
                     "skip the store of 16-byte chunks an action leaves unchanged, so `traffic` (PMC) can be lower")
            if S != 4 else "achieved/frac price the algorithmic bytes 2S^3+3S+1 per step"}


def cpu_baseline(B, S, budget_s=12.0):
    """The reference-dtype torch-CPU port on this host: fp32 (B,1,S,S,S) state, int64 tokens.
    torch's intra-op pool is tried at a few sizes (a 256-thread pool thrashes on these small
    elementwise ops); the fastest is timed for the budget and its size reported as `cores`."""
    import numpy as np
    from oracle import ref_dtype_torch as P

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    rng = np.random.default_rng(0)
    state0 = torch.from_numpy(rng.integers(-2, 3, size=(B, 1, S, S, S)).astype(np.float32))
    acts = torch.from_numpy(rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 1, 3 * S)).astype(np.int64))
    trials = {}
    for nt in sorted({n for n in (4, 8, 16, 32, 64) if n <= avail} | {min(avail, 16)}):
        torch.set_num_threads(nt)
        P.env_step(state0, acts)  # warm-up
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0 or n < 2:
            P.env_step(state0, acts)
            n += 1
        trials[nt] = n / (time.perf_counter() - t0)
    best = max(trials, key=trials.get)
    torch.set_num_threads(best)
    state = state0
    n, t0 = 0, time.perf_counter()
    while True:
        state, done = P.env_step(state, acts)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 5) or n >= 100000:
            break
    # per-game loop (how the reference actually calls it: B=1, k=1), bounded sample
    torch.set_num_threads(1)
    s1 = torch.zeros((1, 1, S, S, S))
    a1 = acts[:1]
    m, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0:
        s1, d1 = P.env_step(s1, a1)
        m += 1
    single = m / (time.perf_counter() - t1)
    # the plain-C int8 restatement (oracle/tg_oracle.c), one thread: what a scalar CPU loop over the
    # build's own int8 layout does -- informational, not the reference's arithmetic dtypes
    c_rate = None
    try:
        from oracle.c_oracle import COracle

        co = COracle()
        st8 = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
        ac8 = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
        co.step_i8(st8, ac8)
        c_n, t2 = 0, time.perf_counter()
        while time.perf_counter() - t2 < 2.0:
            co.step_i8(st8, ac8)
            c_n += 1
        c_rate = round(B * c_n / (time.perf_counter() - t2), 1)
    except Exception as e:  # the C oracle is optional test infrastructure
        c_rate = f"unavailable: {e}"
    return {"value": round(B * n / el, 1), "unit": "steps/s", "cores": best, "kind": "port",
            "c_int8_port_1thread_steps_per_s": c_rate,
            "sample": f"{n} batched steps of the same workload (B={B}, S={S}; fp32 state, int64 tokens, "
                      f"torch-CPU op sequence of get_child_states + zero check) in {el:.1f} s; "
                      f"thread-count sweep (batched steps/s): " + ", ".join(f"{k}:{v:.1f}" for k, v in trials.items()),
            "host_cpus_available": avail, "per_game_loop_steps_per_s": round(single, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2016)
    ap.add_argument("--warmup", type=int, default=224)
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: the K launches are captured in hipGraphs and replayed; eager: K ctypes launches")
    ap.add_argument("--dim", type=int, default=4, help="S of the timed workload (4 = BASELINE config 2)")
    ap.add_argument("--batch", type=int, default=0, help="games per GPU (default: 65536 for S=4, 8192 for S=16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # One process per GPU.  LOCAL_RANK is taken modulo the VISIBLE devices, which is the identity on a full
    # node and maps every rank to device 0 when the launcher exposes one GPU per process (HIP_VISIBLE_DEVICES).
    # Rehearsal only: TG_BENCH_BACKEND=gloo replaces the RCCL control plane (N ranks sharing a 1-GPU box).
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from mat_mul_amd import build as tg_build, shard_range
    from mat_mul_amd.sharding import RankGroup

    group = RankGroup(os.environ.get("TG_BENCH_BACKEND", "nccl"), dev)  # control plane only: barrier + max of the elapsed time
    if group.rank == 0 and tg_build.is_stale():  # normally built by __graft_entry__.build(); self-heal on a fresh tree
        tg_build.build()
    group.barrier()
    from mat_mul_amd import _lib  # noqa: F401  (raises if libtensorgame.so or a symbol is missing: no CPU path)

    S = args.dim
    Bg = args.batch or {4: 65536, 16: 8192, 25: 4096, 9: 32768}.get(S, 4096)
    lo, hi = shard_range(Bg * world, rank, world)  # weak scaling: Bg games per GPU
    B = hi - lo

    res = time_steps(B, S, args.steps, args.warmup, dev, args.mode, seed=0, gid0=lo, sync=group.barrier)
    wall, bad = group.max_over_ranks(res["wall_s"], 0.0 if res["ok"] else 1.0)
    if bad:
        raise SystemExit("bench self-check failed: the state did not return to its start after full cycles")

    if rank == 0:
        total_steps = Bg * world * args.steps
        out = {
            "metric": "env steps/sec (batched games)", "value": round(total_steps / wall, 1), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": f"S={S} int8, batch={Bg} independent games per GPU, one in-place tg_step_i8 "
                                   f"launch per step (BASELINE config {2 if S == 4 else 3})",
                       "S": S, "batch_per_gpu": Bg, "global_batch": Bg * world, "launch": args.mode,
                       "parallelism": f"shard{world} (contiguous game ranges, no collective)",
                       "untimed_graph_warm_replay_steps": res["warm_replay"]},
            "roofline": roofline(B, S, args.steps, res["event_ms"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Bg, S)
        if world == 1 and not args.no_also:
            also = []
            for (s2, b2, k2, label) in [(16, 8192, 512, "BASELINE config 3"), (4, 1 << 17, 1008, "BASELINE config 4 per-GPU share at 8 GPUs"),
                                        (4, 1 << 22, 112, "HBM-streaming batch (268 MB of states)"),
                                        (25, 4096, 208, "config 5 per-GPU step"), (16, 1 << 17, 64, "HBM-streaming batch (537 MB of states)")]:
                if s2 == S and b2 == Bg:
                    continue
                r2 = time_steps(b2, s2, k2, 32, dev, args.mode, seed=1)
                also.append({"workload": f"S={s2} batch={b2} ({label})", "ok": r2["ok"],
                             "value": round(b2 * k2 / r2["wall_s"], 1), "unit": "steps/s",
                             "roofline": roofline(b2, s2, k2, r2["event_ms"])})
            from mat_mul_amd import ops
            # the fused path, labelled separately: K actions per launch, state stays on chip
            for (s2, b2, k2) in [(4, 65536, 7), (16, 8192, 20), (25, 4096, 64)]:
                tok, tgt = ops.gen_demos(b2, s2, k2, dev, seed=2)
                st2 = ops.alloc_states(b2, s2, dev)
                ds = torch.zeros(b2, dtype=torch.int32, device=dev)
                for _ in range(5):
                    ops.step_many(tgt, tok, out=st2, done_step=ds)
                reps = 50
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                fg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(fg, stream=side):  # the launches only: no Python between them
                    for _ in range(reps):
                        ops.step_many(tgt, tok, out=st2, done_step=ds)
                torch.cuda.current_stream(dev).wait_stream(side)
                fg.replay()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fg.replay()
                e1.record()
                torch.cuda.synchronize(dev)
                sec = e0.elapsed_time(e1) * 1e-3 / reps
                nbytes = b2 * (2 * s2 ** 3 + k2 * 3 * s2 + 4)
                also.append({"workload": f"FUSED tg_step_many_i8: S={s2} batch={b2}, K={k2} actions per launch "
                                         f"(bytes per step = (2S^3 + K*3S + 4)/K; not the single-step metric)",
                             # every game ends at zero; a few get there early when the remaining terms cancel
                             "ok": bool(((ds >= 0) & (ds < k2)).all()) and not bool(st2.any()),
                             "value": round(b2 * k2 / sec, 1), "unit": "steps/s",
                             "us_per_launch": round(sec * 1e6, 2), "GBps": round(nbytes / sec / 1e9, 1)})
            # BASELINE config 5's generator (per-GPU share: 4 096 demos, S=25, R=64), with and without the change
            # of basis; bytes = tokens written and re-read + target written; replayed as a hipGraph
            for with_basis in (False, True):
                s2, b2, r2 = 25, 4096, 64
                P = ops.sample_basis(b2, s2, dev, seed=11) if with_basis else None
                tok = torch.empty((b2, r2, 3 * s2), dtype=torch.int8, device=dev)
                tgt = ops.alloc_states(b2, s2, dev)
                ovf = torch.zeros(b2, dtype=torch.uint8, device=dev)
                for _ in range(3):
                    ops.gen_demos(b2, s2, r2, dev, seed=7, basis=P, target=tgt, actions=tok, overflow=ovf)
                reps = 20
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                gg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gg, stream=side):
                    for _ in range(reps):
                        ops.gen_demos(b2, s2, r2, dev, seed=7, basis=P, target=tgt, actions=tok, overflow=ovf)
                torch.cuda.current_stream(dev).wait_stream(side)
                gg.replay()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gg.replay()
                e1.record()
                torch.cuda.synchronize(dev)
                sec = e0.elapsed_time(e1) * 1e-3 / reps
                # self-check: replaying the demo's own actions must bring every target to zero
                _, dstep = ops.step_many(tgt, tok)
                nbytes = b2 * (s2 ** 3 + 2 * 3 * s2 * r2)
                also.append({"workload": f"GENERATOR tg_gen_demos_i8: S={s2} R={r2}, {b2} demos per launch"
                                         f"{' in a random GL(S,Z) basis' if with_basis else ''} (BASELINE config 5 per GPU)",
                             "ok": bool((dstep >= 0).all()) and (with_basis or not bool(ovf.any())),
                             "value": round(b2 / sec, 1), "unit": "demos/s", "us_per_launch": round(sec * 1e6, 2),
                             "GBps": round(nbytes / sec / 1e9, 1), "hbm_frac": round(nbytes / sec / 1e9 / 8000.0, 4),
                             "TMACps": round(b2 * r2 * s2 ** 3 / sec / 1e12, 2)})
            out["also"] = also
            # BASELINE's metric names S=4 and S=16: surface config 3 at the top level as well
            for a3 in also:
                if a3["workload"].startswith("S=16 batch=8192"):
                    out["value_s16"] = a3["value"]
                    out["ms_per_step_s16"] = round(a3["roofline"]["avg_launch_us"] * 1e-3, 6)
                    out["roofline_s16"] = a3["roofline"]
        print(json.dumps(out), flush=True)
    group.close()


if __name__ == "__main__":
    main()
