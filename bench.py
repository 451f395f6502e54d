#!/usr/bin/env python3
"""bench.py -- env steps/s of the tensor-game step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--samples n] [--scaling weak|strong] [--global-batch G]

A "step" is ONE launch of the hot-path kernel (tg_step_i8, in place) over the whole batch of games resident in
HBM.  Workloads:
  N = 1          BASELINE config 2: S=4 int8, 65 536 independent games.
  N > 1          BASELINE config 4 by default (--scaling strong --global-batch 1048576): S=4, 2^20 games sharded
                 over the N GPUs by contiguous global id range (131 072 per GPU at N=8); --scaling weak keeps
                 65 536 games per GPU instead.  No collective on the data path; RCCL carries the barrier and the
                 max-over-ranks of the elapsed time only.
  With --gpus N > 1 and no WORLD_SIZE in the environment, bench.py starts its N ranks itself (child processes
  under torch.distributed.run, before anything touches the GPU) and relays rank 0's JSON line and exit code.

Timing: W untimed warm-up steps, then `samples` timed samples (default 9) of EXACTLY K steps each, every sample
bracketed by a barrier + torch.cuda.synchronize() on both sides; per sample the MAX over ranks is taken, and
`ms_per_step` / `value` are the MEDIAN sample (all samples are listed).  The K steps of a sample are chained
(each consumes the state the previous one wrote), replayed as one hipGraph, and cycle through an action schedule
that returns every game to its start state, so the timed region checks itself.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the step kernel against the 8 TB/s HBM peak.  `avg_launch_us` is measured with HIP events on the
                  launch stream around each of the `samples` replays of the K-launch graph, enqueued back to back
                  (no host sync in between, so the one-off submit latency of a replay overlaps the previous one),
                  median / K.  `achieved`/`frac` price the bytes the step NEEDS (= SURVEY's algorithmic
                  2S^3+3S+1 per game, minus the stores of 16-byte chunks the action leaves unchanged, which the
                  in-place S>=9 kernels skip; computed exactly from the schedule on the device);
                  `frac_algorithmic` prices SURVEY's figure unconditionally, `frac_traffic` the PMC bytes of the
                  committed rocprofv3 passes (profiles/traffic_rNN.json), `copy_ceiling_GBps` a copy kernel of the
                  same footprint measured in the same run;
  cpu_baseline -- the oracle's reference-dtype torch-CPU port (oracle/ref_dtype_torch.py) timed on this box's
                  host cores on a bounded sample of the same workload;
  also         -- the same measurement on the other single-GPU BASELINE configs, SURVEY 8(d)'s dense-state inputs,
                  HBM-streaming batches, the fused step_many and the generator (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6300.0  # the same guide: what a streaming kernel reaches on a footprint far beyond the caches
L2_BYTES = 32 << 20  # 8 XCDs x 4 MiB
INFINITY_CACHE_BYTES = 256 << 20
HBM_REGIME_BYTES = 2 << 30  # from here on the Infinity Cache holds an eighth of the footprint at most


def regime_of(footprint):
    """(regime label, bound) of a launch by the bytes of state it sweeps.  Only footprints of 2 GiB and more are
    priced as HBM streams: up to 1-2 GiB the 256 MiB Infinity Cache still serves a visible share of an in-place
    ping-pong (round 2 measured 6.9-7.8 TB/s of 'HBM' traffic on 256-512 MiB, more than HBM delivers alone)."""
    mib = footprint / 2 ** 20
    if footprint < L2_BYTES:
        return "L2-resident: %.0f MiB of states < 32 MiB of L2; the launch boundary dominates, HBM is not exercised" % mib, "launch"
    if footprint < INFINITY_CACHE_BYTES:
        return "Infinity-Cache-resident: %.0f MiB of states < 256 MiB; HBM is not exercised" % mib, "cache"
    if footprint < HBM_REGIME_BYTES:
        return "cache-assisted: %.0f MiB of states, 1-8x the Infinity Cache, which still serves part of every pass" % mib, "cache"
    return "hbm-streaming: %.0f MiB of states per launch, >= 8x the Infinity Cache" % mib, "hbm"
STEP_KERNEL = {9: "tg::s9_step_kernel<0>"}


def step_kernel_name(S: int, B: int) -> str:
    """The kernel tg_step_i8 launches for aligned int8 states (tg_kernels.hip, launch_apply)."""
    MiB = 1 << 20
    if S == 4:  # non-temporal state loads from 96 MiB of states on, token wait from 384 MiB on; last flag: digit form
        b = B * 64
        return "tg::s4_step_kernel<true, true, true>" if b >= 384 * MiB else (
            "tg::s4_step_kernel<true, false, true>" if b >= 96 * MiB else "tg::s4_step_kernel<false, false, true>")
    if S == 16:  # whole-line stores from 96 MiB of states on, non-temporal state loads in [320 MiB, 1.25 GiB)
        b = B * 4096
        if 320 * MiB <= b < 1280 * MiB:
            return "tg::s16_step_kernel<0, true, true, true>"
        return "tg::s16_step_kernel<0, true, false, true>" if b >= 96 * MiB else "tg::s16_step_kernel<0, false, false, true>"
    if S == 25:  # 15 632-byte game strides: lines in [96 MiB, 1.25 GiB), nt loads in [320 MiB, 1.25 GiB), plain beyond
        b = B * 15632
        if 320 * MiB <= b < 1280 * MiB:
            return "tg::s25_step_kernel<true, true, false>"
        return "tg::s25_step_kernel<true, false, false>" if 96 * MiB <= b < 1280 * MiB else "tg::s25_step_kernel<false, false, false>"
    return STEP_KERNEL.get(S, "tg::slow_kernel<0>")


def bytes_step(S: int) -> int:
    """read state + read action + write state + write done."""
    return 2 * S ** 3 + 3 * S + 1


# ------------------------------------------------------------------------------------------- workloads
def make_demo_schedule(B, S, R, dev, seed, gid0, pad_to=16):
    """Start state + 2R token tensors: the demo's own R actions, then the same actions with u negated (which add
    the terms back).  After R steps every game is zero; after 2R it is back at its start."""
    from mat_mul_amd import ops

    actions, target = ops.gen_demos(B, S, R, dev, seed=seed, game_id_offset=gid0,
                                    target=ops.alloc_states(B, S, dev, pad_to=pad_to, zero=False))
    sched = []
    for k in range(R):
        sched.append(actions[:, k].contiguous())
    for k in range(R):
        a = actions[:, k].clone()
        a[:, :S] = 2 - a[:, :S]  # token = u + 1  ->  -u + 1
        sched.append(a.contiguous())
    return target, sched, {"terminal_after": R}


def make_dense_schedule(B, S, dev, seed):
    """SURVEY.md section 8(d)'s inputs: state entries i.i.d. uniform on {-2..2}, tokens i.i.d. on {0,1,2} with the
    reference's (0.15, 0.7, 0.15) weights, every 97th game set to state = action tensor so that `done` fires.
    Schedule = (a, a with u negated): period 2, every game returns to its start."""
    from mat_mul_amd import ops

    g = torch.Generator(device="cpu").manual_seed(seed)
    state = ops.alloc_states(B, S, dev)
    state.copy_(torch.randint(-2, 3, (B, S, S, S), generator=g, dtype=torch.int8).to(dev))
    probs = torch.tensor([0.15, 0.7, 0.15])
    tok = torch.multinomial(probs, B * 3 * S, replacement=True, generator=g).to(torch.int8).reshape(B, 3 * S).to(dev)
    planted = torch.arange(0, B, 97, device=dev)
    state[planted] = ops.gen_from_factors(tok[planted].unsqueeze(1).contiguous(), S)
    neg = tok.clone()
    neg[:, :S] = 2 - neg[:, :S]
    return state, [tok.contiguous(), neg.contiguous()], {"planted": planted}


def changed_chunks_per_launch(S, sched):
    """Average over the schedule of the number of 16-byte chunks (chunk c = bytes [16c, 16c+16) of a game) that an
    action changes, i.e. that hold an element with u_i v_j w_l != 0.  Exact, computed on the device in slices."""
    N = S ** 3
    nchunk = -(-N // 16)
    total = 0
    for tok in sched:
        B = tok.shape[0]
        sl = max(1, (64 << 20) // (nchunk * 16))
        for b0 in range(0, B, sl):
            t = tok[b0:b0 + sl].to(torch.int16) - 1
            nzu, nzv, nzw = (t[:, :S] != 0), (t[:, S:2 * S] != 0), (t[:, 2 * S:] != 0)
            m = (nzu[:, :, None, None] & nzv[:, None, :, None] & nzw[:, None, None, :]).reshape(t.shape[0], N)
            if nchunk * 16 != N:
                m = torch.nn.functional.pad(m, (0, nchunk * 16 - N))
            total += int(m.reshape(t.shape[0], nchunk, 16).any(dim=2).sum())
    return total / len(sched)


def needed_bytes_per_launch(B, S, sched, inplace=True):
    """Bytes one launch must move: read state + tokens, write `done`, write the state -- except that in place the
    S>=9 kernels do not store 16-byte chunks the action leaves unchanged (the S=4 kernel always stores)."""
    alg = B * bytes_step(S)
    if S == 4 or not inplace or S not in (9, 16, 25):  # (the sizes with a store-skipping in-place kernel)
        return alg
    return B * (S ** 3 + 3 * S + 1) + 16.0 * changed_chunks_per_launch(S, sched)


# ------------------------------------------------------------------------------------------- timing
class StepTimer:
    """W warm-up launches, then `samples` samples of exactly K chained in-place tg_step_i8 launches."""

    def __init__(self, state0, sched, dev, mode="graph", shift=1, pad_to=16):
        from mat_mul_amd import ops

        self.dev, self.mode, self.sched, self.L = dev, mode, sched, len(sched)
        B, S = state0.shape[0], state0.shape[1]
        self.B, self.S = B, S
        self.start = state0
        self.state = ops.alloc_states(B, S, dev, pad_to=pad_to)
        self.state.copy_(state0)
        self.done = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.ovf = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.launch = ops.prepare_step(self.state, sched, self.done, self.ovf, shift=shift)
        self.pos = 0
        self.graphs = {}
        self.lead_in = 8  # launches in front of the K timed ones in the differential event measurement

    def eager(self, n):
        for _ in range(n):
            self.launch(self.pos % self.L)
            self.pos += 1

    def _graph(self, phase, n):
        """hipGraph of n chained launches starting at schedule phase `phase` (captured once per (phase, n))."""
        key = (phase, n)
        if key not in self.graphs:
            cur = torch.cuda.current_stream(self.dev)
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):  # capture records the launches; nothing runs
                for j in range(n):
                    self.launch((phase + j) % self.L)
            cur.wait_stream(side)
            self.graphs[key] = g
        return self.graphs[key]

    def _plan(self, K):
        """The K steps that start at the current position, as [(graph, replays, nodes)]: whole cycles of the
        schedule in one big graph (replayed), the remainder in a per-phase graph."""
        CH = max(self.L, (2048 // self.L) * self.L)
        nfull, rem = divmod(K, CH)
        phase = self.pos % self.L
        plan = []
        if nfull:
            plan.append((self._graph(phase, CH), nfull, CH))
        if rem:
            plan.append((self._graph(phase, rem), 1, rem))
        return plan

    def run_k(self, K):
        """Enqueue exactly K steps (no sync)."""
        if self.mode == "graph":
            for g, reps, _ in self._plan(K):
                for _ in range(reps):
                    g.replay()
            self.pos += K
        else:
            self.eager(K)

    def finish_cycle(self):
        while self.pos % self.L:
            self.eager(1)

    def check(self):
        """Self-check: finish the current cycle of the schedule and compare with the start state."""
        self.finish_cycle()
        torch.cuda.synchronize(self.dev)
        return bool(torch.equal(self.state, self.start)) and not bool(self.ovf.any())

    def _rewind(self, p0):
        """Eager steps until the schedule is back at phase p0 with the state at its start-of-cycle value."""
        self.finish_cycle()
        while self.pos % self.L != p0 % self.L:
            self.eager(1)

    def measure(self, K, W, samples, sync=None):
        """Returns dict(wall_s=[...], event_ms=[...], ok).  Wall samples are sync-bracketed; event samples come
        from a second pass with the replays enqueued back to back."""
        self.eager(W)
        torch.cuda.synchronize(self.dev)
        p0 = self.pos
        # rehearsal (untimed): the same sequence of K-step runs once, so that every graph the timed passes use is
        # captured, instantiated and uploaded (the first replay of a hipGraph uploads it); then back to phase p0
        for _ in range(samples + 1):
            self.run_k(K)
        self._rewind(p0)
        torch.cuda.synchronize(self.dev)
        walls = []
        for _ in range(samples):
            if sync:
                sync()
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            self.run_k(K)
            torch.cuda.synchronize(self.dev)
            walls.append(time.perf_counter() - t0)
        if sync:
            sync()
        self._rewind(p0)
        torch.cuda.synchronize(self.dev)
        # kernel time.  One replay of a hipGraph costs a fixed ~13 us on top of its nodes (ROCm 7.2; measured by
        # tools/graph_overhead_probe.py, profiles/r02_graph_replay_overhead.txt: 17.5 us for 1 node, 62.8 us for 20,
        # 5081 us for 2016), also when replays are enqueued back to back -- runtime submit cost, not kernel time.  So
        # the K launches are timed DIFFERENTIALLY, with HIP events on the launch stream and no host sync in between:
        # T(m lead-in launches + the K launches, one run) - T(m launches, one run), each run paying that fixed cost once.
        m = self.lead_in
        seq = []
        for _ in range(samples):
            seq += [m + K, m]

        def run_seq(evs):
            self.run_k(m)  # the first timed run starts behind work already in flight
            for i, n in enumerate(seq):
                if evs:
                    evs[i].record()
                self.run_k(n)
            if evs:
                evs[len(seq)].record()

        run_seq(None)  # rehearsal: captures / uploads every (phase, length) graph of the sequence
        self._rewind(p0)
        torch.cuda.synchronize(self.dev)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
        run_seq(evs)
        torch.cuda.synchronize(self.dev)
        t = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(seq))]
        event_ms = [t[2 * i] - t[2 * i + 1] for i in range(samples)]          # exactly K launches each
        replay_ms = [t[2 * i] for i in range(samples)]                         # one run of m + K launches, as it is
        return {"wall_s": walls, "event_ms": event_ms, "ok": self.check(), "lead_in": m,
                "run_ms_with_lead_in": replay_ms, "run_ms_lead_in_only": [t[2 * i + 1] for i in range(samples)]}


def copy_ceiling_gbps(B, S, dev, K=64, samples=5, pad_to=16):
    """A copy kernel of the same footprint in the same run: K chained tg_copy_i8 launches (ping-pong between two
    state buffers) replayed as a hipGraph; GB/s = 2 * S^3 * B bytes per launch / median launch time."""
    from mat_mul_amd import ops

    a, b = ops.alloc_states(B, S, dev, pad_to=pad_to), ops.alloc_states(B, S, dev, pad_to=pad_to)
    ops.copy_states(a, b)
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for j in range(K):
            ops.copy_states(a if j % 2 == 0 else b, b if j % 2 == 0 else a)
    cur.wait_stream(side)
    g.replay()
    torch.cuda.synchronize(dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(samples + 1)]
    g.replay()
    for i in range(samples + 1):
        evs[i].record()
        if i < samples:
            g.replay()
    torch.cuda.synchronize(dev)
    us = statistics.median(evs[i].elapsed_time(evs[i + 1]) for i in range(samples)) * 1e3 / K
    return round(2 * S ** 3 * B / (us * 1e-6) / 1e9, 1), round(us, 3)


def measured_traffic(B, S, kernel):
    """HBM bytes per launch of this workload from the committed rocprofv3 PMC passes (profiles/traffic_rNN.json,
    written by profiles/summarize.py), newest round first.  Returned only when the kernel recorded there is the
    kernel the bench runs now: a stale entry (kernel renamed or replaced) is omitted, not quoted."""
    for f in sorted((ROOT / "profiles").glob("traffic_r*.json"), reverse=True):
        try:
            e = json.loads(f.read_text()).get(f"S{S}_B{B}")
        except (OSError, ValueError):
            continue
        if e and "hbm_bytes_per_launch" in e:
            rec = e.get("kernel", "").replace(" ", "")
            if kernel and rec.startswith(kernel.replace(" ", "")):
                return e["hbm_bytes_per_launch"], f.stem.split("_")[-1]
            return None, None
    return None, None


_HBM_COPY = {}


def hbm_copy_ceiling(dev):
    """GB/s of tg_copy_i8 between two 2 GiB buffers (S=4 layout), measured once per run: what an out-of-place stream
    reaches on THIS box when the caches cannot help -- the achievable figure beside the 8 TB/s spec peak."""
    key = str(dev)
    if key not in _HBM_COPY:
        _HBM_COPY[key] = copy_ceiling_gbps(1 << 25, 4, dev, K=6, samples=3)
        torch.cuda.empty_cache()
    return _HBM_COPY[key]


def roofline(B, S, K, event_ms_samples, needed_bytes=None, copy=None, footprint=None, res=None, hbm_copy=None):
    """The roofline object of one workload.  `frac` = needed bytes / median launch time / HBM peak: the bytes the
    launch must move (never more than SURVEY's algorithmic figure), so it cannot be inflated by stores the kernel
    skips; `frac_algorithmic` prices SURVEY's 2S^3+3S+1 unconditionally and CAN exceed what the memory system
    moved; `frac_traffic` prices the PMC-measured bytes."""
    per_launch_s = statistics.median(event_ms_samples) * 1e-3 / K
    alg = B * bytes_step(S)
    need = alg if needed_bytes is None else min(float(needed_bytes), float(alg))
    kernel = step_kernel_name(S, B)
    traffic, tround = measured_traffic(B, S, kernel)
    footprint = footprint if footprint is not None else B * (-(-S ** 3 // 16) * 16)
    regime, bound = regime_of(footprint)
    out = {"bound": bound, "achieved": round(need / per_launch_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(need / per_launch_s / 1e9 / HBM_PEAK_GBS, 4),
           "frac_algorithmic": round(alg / per_launch_s / 1e9 / HBM_PEAK_GBS, 4),
           "frac_traffic": round(traffic / per_launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
           "traffic": traffic, "traffic_round": tround, "kernel": kernel,
           "bytes_per_launch": alg, "needed_bytes_per_launch": int(round(need)),
           "avg_launch_us": round(per_launch_s * 1e6, 3),
           "launch_us_samples": [round(x * 1e3 / K, 3) for x in event_ms_samples],
           "regime": regime,
           "bound_note": {"launch": "priced against the HBM peak as the contract asks, but bound by the dependent-launch "
                                    "boundary (an empty kernel costs 1.55 us per launch on this chip)",
                          "cache": "priced against the HBM peak; the bytes come from the caches, so this is a "
                                   "throughput, not an HBM-roofline fraction",
                          "hbm": "an HBM stream: peak = the 8 TB/s spec figure; hbm_copy_ceiling_GBps = the copy of a "
                                 "2 GiB buffer measured in this run (the guide quotes ~6.3 TB/s achievable)"}[bound],
           "method": "HIP events on the launch stream, no host sync inside the pass: per sample T(run of m lead-in + K "
                     "launches) - T(run of m launches), median over the samples, / K -- the fixed cost of a hipGraph "
                     "replay (~13 us, runtime submit) cancels; frac = needed bytes per launch / that time / peak"}
    if res is not None and "run_ms_with_lead_in" in res:
        m = res["lead_in"]
        out["lead_in_launches"] = m
        out["run_us_median"] = {"m_plus_K_launches": round(statistics.median(res["run_ms_with_lead_in"]) * 1e3, 2),
                                "m_launches": round(statistics.median(res["run_ms_lead_in_only"]) * 1e3, 2)}
        out["graph_replay_fixed_cost_us"] = round(
            statistics.median(res["run_ms_lead_in_only"]) * 1e3 - m * per_launch_s * 1e6, 2)
    if copy is not None:
        out["copy_ceiling_GBps"], out["copy_launch_us"] = copy
        out["frac_of_copy_ceiling"] = round(out["achieved"] / copy[0], 4) if copy[0] else None
    if hbm_copy is not None:
        out["hbm_copy_ceiling_GBps"] = hbm_copy[0]
        out["hbm_achievable_GBps_guide"] = HBM_ACHIEVABLE_GBS
        if bound == "hbm":
            out["frac_of_hbm_copy_ceiling"] = round(out["achieved"] / hbm_copy[0], 4) if hbm_copy[0] else None
    return out


# ------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(B, S, budget_s=12.0, sweep_s=1.0):
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
        while time.perf_counter() - t0 < sweep_s or n < 2:
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
            "sample": f"{n} batched steps of the same workload (B={B}, S={S}) in {el:.1f} s",
            "sample_detail": "fp32 (B,1,S,S,S) state, int64 tokens, the torch-CPU op sequence of get_child_states + zero "
                             "check (oracle/ref_dtype_torch.py); thread-count sweep (batched steps/s): "
                             + ", ".join(f"{k}:{v:.1f}" for k, v in trials.items()),
            "host_cpus_available": avail, "per_game_loop_steps_per_s": round(single, 1)}


# ------------------------------------------------------------------------------------------- N > 1 launch
def self_launch(n_gpus: int, argv) -> int:
    """--gpus N > 1 without a launcher: start the N ranks as child processes (torch.distributed.run, one rank
    per GPU, rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, relay their output, return their
    exit code.  Never re-executes a process that has initialised HIP."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------- the contract line
MAX_LINE_BYTES = 4096   # the driver keeps a tail of stdout; round 3's 28.7 KB line was cut and never parsed
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "samples", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "dry_run")
CONFIG_KEYS = ("workload", "S", "batch_per_gpu", "global_batch", "launch", "parallelism", "batch_rank0", "last_game_id")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "frac_traffic", "traffic", "kernel",
                 "bytes_per_launch", "needed_bytes_per_launch", "avg_launch_us", "copy_ceiling_GBps", "hbm_copy_ceiling_GBps")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "host_cpus_available", "per_game_loop_steps_per_s")
# numeric side objects of the line: S=16 (BASELINE config 3), the sharded extras of an N>1 run, config 4 on one GPU
EXTRA_KEYS = ("value_s16", "ms_per_step_s16", "roofline_s16", "s16_strong", "s4_weak", "s4_strong", "streamed_s4", "cfg4_one_gpu",
              "generator_cfg5", "also_file", "also_ok", "shards")


def lean(obj, keys):
    return {k: obj[k] for k in keys if k in obj}


def contract_line(full) -> str:
    """The ONE line the driver parses, printed LAST: the contract fields, numeric roofline / cpu_baseline objects and
    the S=16 / sharded figures -- no prose, no sample lists.  Everything else (`also`, methods, notes, per-sample
    timings) goes to bench_also.json.  Raises if the line would not fit the driver's stdout tail."""
    out = lean(full, TOP_KEYS)
    out["config"] = lean(full.get("config", {}), CONFIG_KEYS)
    if "roofline" in full:
        out["roofline"] = lean(full["roofline"], ROOFLINE_KEYS)
    if "cpu_baseline" in full:
        out["cpu_baseline"] = lean(full["cpu_baseline"], CPU_KEYS)
    for k in EXTRA_KEYS:
        if k in full:
            v = full[k]
            out[k] = lean(v, ROOFLINE_KEYS) if k.startswith("roofline") else v
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= MAX_LINE_BYTES or "\n" in line:
        raise RuntimeError(f"bench contract line is {len(line)} bytes (limit {MAX_LINE_BYTES}): move fields to bench_also.json")
    return line


def write_side_file(full, also):
    """bench_also.json next to bench.py (and under gpurun_out/ when that exists, so a gpurun call brings it back):
    the full headline object with its prose and sample lists, and the `also` workloads."""
    doc = {"headline": full, "also": also}
    paths = [ROOT / "bench_also.json"]
    if (ROOT / "gpurun_out").is_dir():
        paths.append(ROOT / "gpurun_out" / "bench_also.json")
    written = None
    for p in paths:
        try:
            p.write_text(json.dumps(doc, indent=1) + "\n")
            written = written or p.name
        except OSError:
            pass
    return written


def demo_rank(S):
    return 7 if S == 4 else 8


def sharded_step(group, G, S, dev, mode, K, W, samples, seed=0):
    """One sharded measurement, every rank taking part: G games in total, this rank's contiguous range of global ids,
    W warm-up + `samples` samples of exactly K in-place steps, each bracketed by barrier + synchronize; the wall time
    per sample and the event-timed launch time are MAX-reduced over the ranks."""
    from mat_mul_amd import shard_range

    lo, hi = shard_range(G, group.rank, group.world)
    B = hi - lo
    start, sched, _ = make_demo_schedule(B, S, demo_rank(S), dev, seed, lo)
    tm = StepTimer(start, sched, dev, mode)
    res = tm.measure(K, W, samples, sync=group.barrier)
    my_launch_us = statistics.median(res["event_ms"]) * 1e3 / K
    my_fixed_us = statistics.median(res["run_ms_lead_in_only"]) * 1e3 - res["lead_in"] * my_launch_us
    reduced = group.max_over_ranks(*res["wall_s"], 0.0 if res["ok"] else 1.0, my_launch_us, my_fixed_us)
    walls, bad, launch_us_max, fixed_us_max = list(reduced[:-3]), reduced[-3], reduced[-2], reduced[-1]
    if bad:
        raise SystemExit(f"bench self-check failed (S={S}, {G} games over {group.world} ranks): the state did not return "
                         "to its start after full cycles")
    del tm
    return {"B": B, "G": G, "S": S, "K": K, "walls": walls, "wall": statistics.median(walls), "launch_us_max": launch_us_max,
            "fixed_us_max": fixed_us_max, "res": res, "sched": sched}


def shard_figures(m):
    """The numeric summary of one sharded_step for the line: whole-job steps/s on the wall clock and on kernel time."""
    return {"global_batch": m["G"], "batch_per_gpu": m["G"] // max(1, m.get("world", 1)),
            "value": round(m["G"] * m["K"] / m["wall"], 1), "ms_per_step": round(m["wall"] * 1e3 / m["K"], 6),
            "launch_us": round(m["launch_us_max"], 3), "event_steps_per_s": round(m["G"] / (m["launch_us_max"] * 1e-6), 1)}


def streamed_time(b2, s2, k2, dev, seed=4, gid0=0, reps=5):
    """tg_step_stream_i8 on b2 games: K steps in ONE resident launch, the actions consumed step by step.  Returns
    (seconds per launch, ok, games per wavefront, 'ready' | 'rounds')."""
    from mat_mul_amd import ops

    r2 = demo_rank(s2)
    tok, tgt = ops.gen_demos(b2, s2, r2, dev, seed=seed, game_id_offset=gid0)
    cyc = torch.cat([tok, tok], dim=1)
    cyc[:, r2:, :s2] = 2 - cyc[:, r2:, :s2]                  # the same terms with u negated: period 2 r2
    acts = cyc.permute(1, 0, 2).contiguous().repeat(k2 // (2 * r2), 1, 1)   # (K, B, 3S) step-major
    st2 = ops.alloc_states(b2, s2, dev)
    st2.copy_(tgt)
    dn = torch.empty((k2, b2), dtype=torch.uint8, device=dev)
    if b2 <= ops.step_stream_capacity(s2, dev):
        n_units, gpw = ops.step_stream_layout(b2, s2, dev)
        ready = torch.ones(k2, dtype=torch.int32, device=dev)    # pre-set: the producer is never the bottleneck
    else:  # beyond the resident batch: no ready words, the units (S=4: 64 games each, S=16/25: one game) run in rounds
        n_units, gpw, ready = (-(-b2 // 64), 64, None) if s2 == 4 else (b2, 1, None)
    prog = torch.zeros(n_units, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: ops.step_stream(st2, acts, done=dn, ready=ready, progress=prog, status=status)
    fn()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ok = bool(torch.equal(st2, tgt)) and int(status[0]) == 0 and bool((prog == k2).all()) \
        and bool(dn[r2 - 1].all()) and not bool(dn[0].all())
    return statistics.median(ts), ok, gpw, ("ready" if ready is not None else "rounds")


def sharded_extras(group, world, dev, mode, K, headline):
    """N > 1 only, after the headline's timed region, every rank taking part: BASELINE config 3 sharded weak (8 192 games
    per GPU) and strong (8 192 in total), config 4 strong (2^20 games in total) with its single-GPU denominator and the
    resident stepper on the same share, measured in the same run."""
    from mat_mul_amd import shard_range

    rank = group.rank
    Kx = max(14, min(K, 512))
    out = {}
    m = sharded_step(group, 8192 * world, 16, dev, mode, Kx, 16, 5, seed=1)
    m["world"] = world
    if rank == 0:
        f = shard_figures(m)
        out["value_s16"], out["ms_per_step_s16"] = f["value"], f["ms_per_step"]
        out["roofline_s16"] = roofline(m["B"], 16, Kx, m["res"]["event_ms"], needed_bytes_per_launch(m["B"], 16, m["sched"]))
    m = sharded_step(group, 8192, 16, dev, mode, Kx, 16, 5, seed=1)
    m["world"] = world
    if rank == 0:
        out["s16_strong"] = shard_figures(m)
    G4 = 1 << 20
    if headline["S"] == 4 and headline["G"] == G4:     # --scaling strong: config 4 IS the headline; add the weak figure
        m = dict(headline, world=world)
        Kx4 = headline["K"]
        mw = sharded_step(group, 65536 * world, 4, dev, mode, Kx, 16, 5, seed=0)
        mw["world"] = world
        if rank == 0:
            out["s4_weak"] = shard_figures(mw)
    else:
        m = sharded_step(group, G4, 4, dev, mode, Kx, 16, 5, seed=0)
        m["world"] = world
        Kx4 = Kx
    # the resident stepper on this rank's share of config 4 (K steps in one launch); MAX over ranks
    lo, hi = shard_range(G4, rank, world)
    ks = 14 * max(1, min(Kx, 504) // 14)
    group.barrier()
    sec, ok, gpw, how = streamed_time(hi - lo, 4, ks, dev, gid0=lo)
    (sec_max, bad) = group.max_over_ranks(sec, 0.0 if ok else 1.0)
    if bad:
        raise SystemExit("bench self-check failed: streamed stepper on the config-4 share")
    if rank == 0:
        f = shard_figures(m)
        # the same GLOBAL batch on ONE GPU (rank 0 alone): the denominator of the strong-scaling speed-up, same run
        s1, sc1, _ = make_demo_schedule(G4, 4, 7, dev, 0, 0)
        t1 = StepTimer(s1, sc1, dev, mode)
        k1 = max(14, min(Kx, 112))
        r1 = t1.measure(k1, 14, 5)
        if not r1["ok"]:
            raise SystemExit("bench self-check failed: config 4 on one GPU")
        one_us = statistics.median(r1["event_ms"]) * 1e3 / k1
        fixed1 = statistics.median(r1["run_ms_lead_in_only"]) * 1e3 - r1["lead_in"] * one_us
        del t1, s1, sc1
        # what a short wall clock hides: every timed sample is ONE hipGraph replay whose fixed submit cost (~13 us on
        # ROCm 7.2) is paid once per sample whatever the batch; both speed-ups are reported
        f.update({"one_gpu_launch_us": round(one_us, 3), "speedup_event": round(one_us / m["launch_us_max"], 3),
                  "speedup_wall": round((Kx4 * one_us + fixed1) / (m["wall"] * 1e6), 3), "ideal": world})
        out["s4_strong"] = f
        sec1, ok1, _, how1 = streamed_time(G4, 4, 112, dev)
        out["streamed_s4"] = {"global_batch": G4, "share_us_per_step": round(sec_max / ks * 1e6, 3), "share_mode": how,
                              "one_gpu_us_per_step": round(sec1 / 112 * 1e6, 3), "one_gpu_mode": how1, "ok": bool(ok1),
                              "speedup": round((sec1 / 112) / (sec_max / ks), 3),
                              "value": round(G4 / (sec_max / ks), 1)}
        torch.cuda.empty_cache()
    group.barrier()
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2016)
    ap.add_argument("--warmup", type=int, default=224)
    ap.add_argument("--samples", type=int, default=9, help="timed samples of exactly --steps steps each (median reported)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: the K launches of a sample are one hipGraph replay; eager: K ctypes launches")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak (default) = --batch games per GPU, the N=1 workload on every GPU; strong = --global-batch "
                         "games in total (BASELINE config 4).  The other one is measured as well and reported beside it")
    ap.add_argument("--global-batch", type=int, default=1 << 20, help="games in total under --scaling strong")
    ap.add_argument("--dim", type=int, default=4, help="S of the timed workload (4 = BASELINE config 2 / 4)")
    ap.add_argument("--batch", type=int, default=0, help="games per GPU (default: 65536 for S=4, 8192 for S=16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--print-also", action="store_true", help="print the `also` entries as separate lines BEFORE the contract line")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous, shard arithmetic and the JSON line only -- no GPU work (CPU test of the N>1 path)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.samples < 1 or args.steps < 1:
        raise SystemExit("--samples and --steps must be >= 1")

    from mat_mul_amd import shard_range
    from mat_mul_amd.sharding import RankGroup

    S = args.dim
    scaling = args.scaling if world > 1 else "weak"
    per_gpu_default = {4: 65536, 16: 8192, 25: 4096, 9: 32768}.get(S, 4096)
    if scaling == "strong":
        G = args.global_batch
    else:
        G = (args.batch or per_gpu_default) * world
    lo, hi = shard_range(G, rank, world)  # contiguous global game ids of this rank
    B = hi - lo
    if scaling == "strong":
        cfg = f"BASELINE config 4: S={S} int8, {G} games in total sharded over {world} GPUs ({G // world} per GPU)"
    else:
        cfg = (f"BASELINE config {2 if S == 4 else 3 if S == 16 else '-'}: S={S} int8, batch={G // world} independent "
               f"games per GPU")
    head = {"metric": "env steps/sec (batched games)", "value": None, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "samples": args.samples, "ms_per_step": None,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": cfg + "; one in-place tg_step_i8 launch per step", "S": S, "batch_per_gpu": G // world,
                       "global_batch": G, "launch": args.mode,
                       "parallelism": f"shard{world} (contiguous game ranges, no collective)"}}

    if args.dry_run:
        group = RankGroup("gloo")
        group.barrier()
        (tot,) = group.max_over_ranks(float(hi))
        if rank == 0:
            head["dry_run"] = True
            head["config"].update({"batch_rank0": B, "last_game_id": int(tot)})
            if world > 1:  # the shard arithmetic of the extras an N>1 run measures (sharded_extras)
                head["shards"] = {}
                for key, (g2, s2) in {"s16_weak": (8192 * world, 16), "s16_strong": (8192, 16), "s4_strong": (1 << 20, 4),
                                      "s4_weak": (65536 * world, 4)}.items():
                    l2, h2 = shard_range(g2, 0, world)
                    head["shards"][key] = {"S": s2, "global_batch": g2, "batch_rank0": h2 - l2}
            print(contract_line(head), flush=True)
        group.close()
        return 0

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # One process per GPU.  LOCAL_RANK is taken modulo the VISIBLE devices, which is the identity on a full
    # node and maps every rank to device 0 when the launcher exposes one GPU per process (HIP_VISIBLE_DEVICES).
    # Rehearsal only: TG_BENCH_BACKEND=gloo replaces the RCCL control plane (N ranks sharing a 1-GPU box).
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from mat_mul_amd import build as tg_build

    group = RankGroup(os.environ.get("TG_BENCH_BACKEND", "nccl"), dev)  # control plane only: barrier + max of the elapsed time
    if group.rank == 0 and tg_build.is_stale():  # normally built by __graft_entry__.build(); self-heal on a fresh tree
        tg_build.build()
    group.barrier()
    from mat_mul_amd import _lib  # noqa: F401  (raises if libtensorgame.so or a symbol is missing: no CPU path)

    m = sharded_step(group, G, S, dev, args.mode, args.steps, args.warmup, args.samples)
    res, sched, walls, wall = m["res"], m["sched"], m["walls"], m["wall"]
    extras = sharded_extras(group, world, dev, args.mode, args.steps, m) if world > 1 and not args.no_also else {}

    if rank == 0:
        need = needed_bytes_per_launch(B, S, sched)
        copy = copy_ceiling_gbps(B, S, dev)
        hbm_copy = hbm_copy_ceiling(dev) if world == 1 and not args.no_also else None
        out = dict(head)
        out.update({"value": round(G * args.steps / wall, 1), "ms_per_step": round(wall * 1e3 / args.steps, 6)})
        out["config"] = dict(head["config"], timing=f"{args.samples} samples of exactly {args.steps} steps, each bracketed by "
                             f"barrier + synchronize, max over ranks per sample, median over samples",
                             wall_us_per_sample=[round(w * 1e6, 1) for w in walls])
        out["roofline"] = roofline(B, S, args.steps, res["event_ms"], need, copy, res=res, hbm_copy=hbm_copy)
        if world > 1:
            out["per_rank"] = {"event_launch_us_max_over_ranks": round(m["launch_us_max"], 3),
                               "graph_replay_fixed_cost_us_max_over_ranks": round(m["fixed_us_max"], 2),
                               "event_steps_per_s": round(G / (m["launch_us_max"] * 1e-6), 1),
                               "wall_us_per_step": round(wall * 1e6 / args.steps, 3)}
            out.update(extras)
        if not args.no_cpu_baseline:  # rank 0, after every timed region; a shorter sample when other ranks wait for it
            out["cpu_baseline"] = cpu_baseline(G // world, S, budget_s=12.0 if world == 1 else 5.0, sweep_s=1.0 if world == 1 else 0.4)
        also = []
        if world == 1 and not args.no_also:
            also = also_lines(S, G, dev, args.mode, hbm_copy)
            # BASELINE's metric names S=4 and S=16: surface config 3 at the top level as well
            for a3 in also:
                wl = a3["workload"]
                if wl.startswith("S=16 batch=8192 (BASELINE config 3)"):
                    out["value_s16"] = a3["value"]
                    out["ms_per_step_s16"] = round(a3["roofline"]["avg_launch_us"] * 1e-3, 6)
                    out["roofline_s16"] = a3["roofline"]
                elif wl.startswith("S=4 batch=131072 (BASELINE config 4 per-GPU share"):
                    out.setdefault("cfg4_one_gpu", {})["share_of_8_launch_us"] = a3["roofline"]["avg_launch_us"]
                elif wl.startswith("S=4 batch=1048576 (BASELINE config 4 on ONE GPU"):
                    out.setdefault("cfg4_one_gpu", {})["whole_launch_us"] = a3["roofline"]["avg_launch_us"]
                elif wl.startswith("STREAMED tg_step_stream_i8: S=4 batch=131072"):
                    out.setdefault("cfg4_one_gpu", {})["streamed_share_of_8_us_per_step"] = a3["us_per_step"]
                elif wl.startswith("STREAMED tg_step_stream_i8: S=4 batch=1048576"):
                    out.setdefault("cfg4_one_gpu", {})["streamed_whole_us_per_step"] = a3["us_per_step"]
                elif wl.startswith("GENERATOR tg_gen_demos_i8") and "basis" not in wl:
                    out["generator_cfg5"] = {"us_per_launch": a3["us_per_launch"], "demos_per_s": a3["value"],
                                             "bound": a3.get("bound"), "valu_issue_frac": a3.get("valu_issue_frac")}
            c4 = out.get("cfg4_one_gpu", {})
            if "share_of_8_launch_us" in c4 and "whole_launch_us" in c4:  # what 8 GPUs can reach on kernel time
                c4["predicted_speedup_8gpu_event"] = round(c4["whole_launch_us"] / c4["share_of_8_launch_us"], 3)
            if "streamed_share_of_8_us_per_step" in c4 and "streamed_whole_us_per_step" in c4:
                c4["predicted_speedup_8gpu_streamed"] = round(c4["streamed_whole_us_per_step"] / c4["streamed_share_of_8_us_per_step"], 3)
            out["also_ok"] = {"entries": len(also), "failed": [a["workload"][:60] for a in also if not a.get("ok", True)]}
        if also or world > 1:
            out["also_file"] = write_side_file(out, also)
        if args.print_also:
            for a in also:
                print(json.dumps(a), flush=True)
        print(contract_line(out), flush=True)   # LAST line of stdout
    group.barrier()
    group.close()
    return 0


def also_lines(S_main, B_main, dev, mode, hbm_copy=None):
    """The other single-GPU workloads (informational; every line carries its own self-check)."""
    from mat_mul_amd import ops

    also = []
    for (s2, b2, k2, label, kind) in [
            (16, 8192, 512, "BASELINE config 3", "demo"),
            (4, 65536, 1008, "SURVEY 8(d) dense inputs: uniform {-2..2} states, every 97th game terminal", "dense"),
            (16, 8192, 512, "SURVEY 8(d) dense inputs: uniform {-2..2} states, every 97th game terminal", "dense"),
            (4, 1 << 17, 1008, "BASELINE config 4 per-GPU share at 8 GPUs", "demo"),
            (4, 1 << 20, 112, "BASELINE config 4 on ONE GPU (67 MB of states)", "demo"),
            (4, 1 << 22, 112, "cache-assisted batch (256 MiB of states = the Infinity Cache)", "demo"),
            (25, 4096, 208, "config 5 per-GPU step", "demo"),
            (16, 1 << 17, 64, "cache-assisted batch (512 MiB of states)", "demo"),
            (25, 1 << 15, 64, "BASELINE config 5 whole on ONE GPU (488 MiB of states, cache-assisted)", "demo"),
            (4, 1 << 25, 16, "HBM stream: 2 GiB of states", "demo"),
            (16, 1 << 19, 16, "HBM stream: 2 GiB of states", "demo"),
            (25, 139264, 16, "HBM stream: 2.0 GiB of states", "demo")]:
        if kind == "demo" and s2 == S_main and b2 == B_main:
            continue
        if kind == "demo":
            st, sc, info = make_demo_schedule(b2, s2, 7 if s2 == 4 else 8, dev, 1, 0)
        else:
            st, sc, info = make_dense_schedule(b2, s2, dev, 5)
        tm = StepTimer(st, sc, dev, mode)
        extra_ok = True
        if kind == "dense":  # `done` fires exactly on the planted games after the first action
            tm.eager(1)
            torch.cuda.synchronize(dev)
            want = torch.zeros(b2, dtype=torch.uint8, device=dev)
            want[info["planted"]] = 1
            extra_ok = bool(torch.equal(tm.done, want))
            tm.eager(1)
        huge = b2 * s2 ** 3 >= (1 << 30)
        r2 = tm.measure(k2, 8 if huge else 32, 3 if huge else 5)
        w2 = statistics.median(r2["wall_s"])
        need2 = needed_bytes_per_launch(b2, s2, sc)
        del tm
        also.append({"workload": f"S={s2} batch={b2} ({label})", "ok": r2["ok"] and extra_ok,
                     "value": round(b2 * k2 / w2, 1), "unit": "steps/s", "steps": k2,
                     "roofline": roofline(b2, s2, k2, r2["event_ms"], need2,
                                          copy_ceiling_gbps(b2, s2, dev, K=6 if huge else (16 if b2 * s2 ** 3 > (1 << 27) else 64),
                                                            samples=3 if huge else 5), hbm_copy=hbm_copy)})
        del st, sc
        torch.cuda.empty_cache()
    # the fused path, labelled separately: K actions per launch, state stays on chip
    for (s2, b2, k2) in [(4, 65536, 7), (16, 8192, 20), (25, 4096, 64)]:
        tok, tgt = ops.gen_demos(b2, s2, k2, dev, seed=2)
        st2 = ops.alloc_states(b2, s2, dev)
        ds = torch.zeros(b2, dtype=torch.int32, device=dev)
        sec = graph_time(lambda: ops.step_many(tgt, tok, out=st2, done_step=ds), dev, reps=50)
        nbytes = b2 * (2 * s2 ** 3 + k2 * 3 * s2 + 4)
        also.append({"workload": f"FUSED tg_step_many_i8: S={s2} batch={b2}, K={k2} actions per launch "
                                 f"(bytes per step = (2S^3 + K*3S + 4)/K; not the single-step metric)",
                     # every game ends at zero; a few get there early when the remaining terms cancel
                     "ok": bool(((ds >= 0) & (ds < k2)).all()) and not bool(st2.any()),
                     "value": round(b2 * k2 / sec, 1), "unit": "steps/s",
                     "us_per_launch": round(sec * 1e6, 2), "GBps": round(nbytes / sec / 1e9, 1)})
    # the streamed stepper (its own entry and metric): K steps in ONE resident launch, the state stays in registers,
    # per step poll + 12 token bytes in + state and done written through + a progress word per wavefront
    for (s2, b2, k2) in [(4, 65536, 1008), (4, 131072, 504), (16, 8192, 512), (25, 4096, 256), (4, 1 << 20, 112),
                         (25, 32768, 64)]:  # (the last two: BASELINE configs 4 and 5 whole on one GPU, units in rounds)
        sec, ok, gpw, how = streamed_time(b2, s2, k2, dev)
        # per step: tokens in, done out, and the state written through: the whole game once per block of steps
        if s2 == 4:   # the state leaves once per block of D steps (D = 8 for 16 or 64 games per wavefront, 4 for 32)
            moved = b2 * (3 * s2 + 1) + b2 * s2 ** 3 / {16: 8, 32: 4, 64: 8}.get(gpw, 2)
        else:         # S=16 / S=25: whole games once per block of 8 steps
            moved = b2 * (3 * s2 + 1) + b2 * s2 ** 3 / 8
        also.append({"workload": f"STREAMED tg_step_stream_i8: S={s2} batch={b2}, K={k2} steps in ONE launch, actions "
                                 f"consumed step by step ({'ready words pre-set' if how == 'ready' else 'no ready words: beyond the resident batch, the units run in rounds'}), "
                                 f"progress published per wavefront; not the single-step metric",
                     "ok": ok, "value": round(b2 * k2 / sec, 1), "unit": "steps/s", "us_per_step": round(sec / k2 * 1e6, 3),
                     "games_per_wavefront": gpw, "GBps_moved": round(moved * k2 / sec / 1e9, 1),
                     "frac_of_hbm_peak_moved_bytes": round(moved * k2 / sec / 1e9 / HBM_PEAK_GBS, 4)})
    # BASELINE config 5's generator (per-GPU share: 4 096 demos, S=25, R=64), with and without the change
    # of basis; bytes = target + tokens written (S^3 + 3SR per demo); replayed as a hipGraph
    for with_basis in (False, True):
        s2, b2, r2 = 25, 4096, 64
        P = ops.sample_basis(b2, s2, dev, seed=11) if with_basis else None
        tok = torch.empty((b2, r2, 3 * s2), dtype=torch.int8, device=dev)
        tgt = ops.alloc_states(b2, s2, dev)
        ovf = torch.zeros(b2, dtype=torch.uint8, device=dev)
        sec = graph_time(lambda: ops.gen_demos(b2, s2, r2, dev, seed=7, basis=P, target=tgt, actions=tok, overflow=ovf),
                         dev, reps=20)
        # self-check: replaying the demo's own actions must bring every target to zero
        _, dstep = ops.step_many(tgt, tok)
        nbytes = b2 * (s2 ** 3 + 3 * s2 * r2)
        also.append({"workload": f"GENERATOR tg_gen_demos_i8: S={s2} R={r2}, {b2} demos per launch"
                                 f"{' in a random GL(S,Z) basis' if with_basis else ''} (BASELINE config 5 per GPU)",
                     "ok": bool((dstep >= 0).all()) and (with_basis or not bool(ovf.any())),
                     "value": round(b2 / sec, 1), "unit": "demos/s", "us_per_launch": round(sec * 1e6, 2),
                     "GBps": round(nbytes / sec / 1e9, 1), "hbm_frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4),
                     "TMACps": round(b2 * r2 * s2 ** 3 / sec / 1e12, 2),
                     "int8_mfma_frac": round(2 * b2 * r2 * s2 ** 3 / sec / 5.0e15, 4)})
        # which bound binds: neither HBM (hbm_frac) nor the matrix cores (int8_mfma_frac) -- the vector ALU's
        # instruction issue: Philox, draw evaluation, the byte products and the packing around the MFMAs
        insts, rnd = generator_valu_instructions(with_basis)
        also[-1]["bound"] = "valu"
        if insts:
            also[-1]["valu_issue_frac"] = round(insts * VALU_ISSUE_CYCLES / (SIMDS * SHADER_CLOCK_HZ * sec), 4)
            also[-1]["valu_instructions_per_launch"] = int(insts)
            also[-1]["valu_counter_round"] = rnd
            also[-1]["valu_note"] = ("SQ_INSTS_VALU of the committed PMC pass x 4 issue cycles / (1024 SIMDs x 2.4 GHz x the time "
                                     "measured here); issue alone (measured 4.3 cycles per VOP3 instruction, "
                                     "profiles/r02_issue_rates.txt) would read %.2f" % (insts * 4.3 / (SIMDS * SHADER_CLOCK_HZ * sec)))
    # the step that loads only what the action touches (tg_step_tracked_i8: the count of non-zero entries is carried, so the
    # zero test needs no pass over the game): against tg_step_i8 on the same schedule, both by hipGraph replay of whole
    # cycles (the state and the count are back at their start after every replay: checked)
    for s2, b2, reps in ((16, 8192, 64), (16, 1 << 17, 32), (16, 1 << 19, 16), (25, 4096, 64), (25, 1 << 15, 32), (25, 139264, 16)):
        start, sched, _ = make_demo_schedule(b2, s2, 8, dev, 1, 0)
        L = len(sched)
        st_full, st_tr = ops.alloc_states(b2, s2, dev), ops.alloc_states(b2, s2, dev)
        st_full.copy_(start)
        st_tr.copy_(start)
        nnz0 = ops.done(start, want_nnz=True)[1]
        nnz = nnz0.clone()
        dn = torch.zeros(b2, dtype=torch.uint8, device=dev)
        pos = [0]

        def full():
            ops.step(st_full, sched[pos[0] % L], out=st_full, done=dn)
            pos[0] += 1

        def tracked():
            ops.step_tracked(st_tr, sched[pos[0] % L], nnz, done=dn)
            pos[0] += 1

        pos[0] = 0
        t_full = graph_time(full, dev, reps=reps)
        while pos[0] % L:  # (graph_time's three eager warm-up calls left the schedule mid-cycle: finish it)
            full()
        pos[0] = 0
        t_tr = graph_time(tracked, dev, reps=reps)
        while pos[0] % L:
            tracked()
        torch.cuda.synchronize(dev)
        ok = bool(torch.equal(st_tr, start)) and bool(torch.equal(nnz, nnz0)) and bool(torch.equal(st_full, start))
        also.append({"workload": f"TRACKED tg_step_tracked_i8: S={s2} batch={b2} ({b2 * (-(-s2 ** 3 // 16) * 16) >> 20} MiB of states), in "
                                 "place, the count of non-zero entries carried: only the rows the action touches are loaded",
                     "ok": ok, "value": round(b2 / t_tr, 1), "unit": "steps/s", "us_per_launch": round(t_tr * 1e6, 2),
                     "us_tg_step_i8": round(t_full * 1e6, 2), "gain": round(t_full / t_tr, 3)})
        del st_full, st_tr, start, sched
        torch.cuda.empty_cache()
    # get_child_states with k > 1 (act.py:266-275, the shape MCTS expansion calls): k children per parent in one launch;
    # bytes = parent read + k x (child written + tokens read + done + changed)
    for s2, b2 in ((4, 65536), (4, 1 << 20), (9, 32768), (16, 8192), (25, 4096)):
        k2 = 8
        tok, tgt = ops.gen_demos(b2, s2, k2, dev, seed=5)
        kids = ops.alloc_states(b2 * k2, s2, dev).unflatten(0, (b2, k2))
        kd = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        kc = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        sec = graph_time(lambda: ops.expand(tgt, tok, out=kids, done=kd, changed=kc), dev, reps=20 if b2 * s2 ** 3 < (32 << 20) else 5)
        st, dn = ops.step(tgt, tok[:, k2 - 1].contiguous())      # self-check: child c == one step with action c
        ok = bool(torch.equal(kids[:, k2 - 1], st)) and bool(torch.equal(kd[:, k2 - 1], dn))
        nbytes = b2 * (s2 ** 3 + k2 * (s2 ** 3 + 3 * s2 + 2))
        also.append({"workload": f"EXPAND tg_expand_i8: S={s2} batch={b2}, k={k2} children per parent in one launch "
                                 f"(get_child_states with k > 1); a write stream: {k2 * b2 * s2 ** 3 / 1e6:.0f} MB of children",
                     "ok": ok, "value": round(b2 * k2 / sec, 1), "unit": "children/s", "us_per_launch": round(sec * 1e6, 2),
                     "GBps": round(nbytes / sec / 1e9, 1), "hbm_frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4)})
        del kids, kd, kc, tok, tgt
    also += fused_lines(dev)
    return also


def fused_lines(dev, batches=(65536, 1 << 20)):
    """SURVEY N1 / N2 fused entries against the two calls they replace (S=4: the shape MCTS expansion runs at; S=16: the
    metric's other size, keys only)."""
    from mat_mul_amd import ops

    also = []
    for s2, b2, k2, reps in ((16, 8192, 8, 5), (25, 4096, 8, 5)):
        tok, tgt = ops.gen_demos(b2, s2, k2, dev, seed=6)
        kids = ops.alloc_states(b2 * k2, s2, dev).unflatten(0, (b2, k2))
        kd = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        kc = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        keys = torch.zeros((b2, k2), dtype=torch.int64, device=dev)
        fused = graph_time(lambda: ops.expand(tgt, tok, out=kids, done=kd, changed=kc, keys=keys), dev, reps=reps)
        kf = keys.clone()
        two = graph_time(lambda: (ops.expand(tgt, tok, out=kids, done=kd, changed=kc),
                                  keys.view(-1).copy_(ops.state_hash(kids.flatten(0, 1)))), dev, reps=reps)
        also.append({"workload": f"N2 FUSED tg_expand_keyed_i8: S={s2} batch={b2}, k={k2}: children + their 64-bit keys in one "
                                 f"launch (extend_tree's state_to_str per child, act.py:188-190)",
                     "ok": bool(torch.equal(keys, kf)), "us_per_launch": round(fused * 1e6, 2),
                     "us_expand_then_hash": round(two * 1e6, 2), "gain": round(two / fused, 3)})
        del kids, kd, kc, keys, tok, tgt
        torch.cuda.empty_cache()
    # N1 at S=16: tg_step_emit is one kernel while its output stays below 128 MiB (1 024 games here), two launches inside the
    # call beyond (8 192 games: the frames' write stream is the whole cost); beside it the step and the frames alone
    for b2 in (1024, 8192):
        s2, T = 16, 4
        tok, tgt = ops.gen_demos(b2, s2, 2, dev, seed=6)
        a0 = tok[:, 0].contiguous()
        ring = ops.alloc_ring(b2, s2, T, dev)
        for f in range(T):
            ring[:, f].copy_(tgt)
        dn = torch.zeros(b2, dtype=torch.uint8, device=dev)
        sc = torch.empty((b2, 1), dtype=torch.float32, device=dev)
        x = torch.empty((b2, T, s2, s2, s2), dtype=torch.float16, device=dev)
        both = graph_time(lambda: ops.step_emit(ring, 0, a0, 1.0, torch.float16, out=x, scalars=sc, done=dn), dev, reps=10)
        step_only = graph_time(lambda: ops.step(ring[:, 0], a0, out=ring[:, 1], done=dn), dev, reps=10)
        emit_only = graph_time(lambda: ops.emit_frames(ring, 1, 1.0, torch.float16, out=x, scalars=sc), dev, reps=10)
        nbytes = b2 * (s2 ** 3 * (T - 1) + 3 * s2 + s2 ** 3 + 1 + T * s2 ** 3 * 2 + 4)
        pair = graph_time(lambda: (ops.step(ring[:, 0], a0, out=ring[:, 1], done=dn),
                                   ops.emit_frames(ring, 1, 1.0, torch.float16, out=x, scalars=sc)), dev, reps=10)
        ops.step_emit(ring, 0, a0, 1.0, torch.float16, out=x, scalars=sc, done=dn)
        xf = x.clone()
        ops.step(ring[:, 0], a0, out=ring[:, 1], done=dn)
        x2, _ = ops.emit_frames(ring, 1, 1.0, torch.float16)
        one_kernel = T * b2 * s2 ** 3 * 2 < (128 << 20)
        also.append({"workload": f"N1 {'FUSED ' if one_kernel else ''}tg_step_emit: S={s2} batch={b2} T={T} float16 ("
                                 f"{'one kernel' if one_kernel else 'two launches inside the call'}; "
                                 f"{T * b2 * s2 ** 3 * 2 / 1e6:.0f} MB of output)",
                     "ok": bool(torch.equal(xf, x2)), "us_per_launch": round(both * 1e6, 2), "us_step_alone": round(step_only * 1e6, 2),
                     "us_emit_frames_alone": round(emit_only * 1e6, 2),
                     "us_step_then_emit_frames": round(pair * 1e6, 2), "gain": round(pair / both, 3),
                     "GBps": round(nbytes / both / 1e9, 1), "hbm_frac": round(nbytes / both / 1e9 / HBM_PEAK_GBS, 4)})
        del x, ring, tok, tgt
        torch.cuda.empty_cache()
    for b2 in batches:
        s2, T, k2 = 4, 4, 8
        reps = 20 if b2 <= 65536 else 5
        tok, tgt = ops.gen_demos(b2, s2, k2, dev, seed=6)
        a0 = tok[:, 0].contiguous()
        ring = ops.alloc_ring(b2, s2, T, dev)
        for f in range(T):
            ring[:, f].copy_(tgt)
        dn = torch.zeros(b2, dtype=torch.uint8, device=dev)
        sc = torch.empty((b2, 1), dtype=torch.float32, device=dev)
        for dt, w in ((torch.float16, 2), (torch.float32, 4)):
            x = torch.empty((b2, T, s2, s2, s2), dtype=dt, device=dev)
            fused = graph_time(lambda: ops.step_emit(ring, 0, a0, 1.0, dt, out=x, scalars=sc, done=dn), dev, reps=reps)
            xf = x.clone()
            two = graph_time(lambda: (ops.step(ring[:, 0], a0, out=ring[:, 1], done=dn),
                                      ops.emit_frames(ring, 1, 1.0, dt, out=x, scalars=sc)), dev, reps=reps)
            nbytes = b2 * (s2 ** 3 * (T - 1) + 3 * s2 + s2 ** 3 + 1 + T * s2 ** 3 * w + 4)
            also.append({"workload": f"N1 FUSED tg_step_emit: S={s2} batch={b2} T={T} {str(dt)[6:]}: one step on the history "
                                     f"ring + the (B,T,S,S,S) model input of the new state in one launch",
                         "ok": bool(torch.equal(x, xf)), "us_per_launch": round(fused * 1e6, 2),
                         "us_step_then_emit_frames": round(two * 1e6, 2), "gain": round(two / fused, 3),
                         "GBps": round(nbytes / fused / 1e9, 1), "hbm_frac": round(nbytes / fused / 1e9 / HBM_PEAK_GBS, 4)})
            del x, xf
        kids = ops.alloc_states(b2 * k2, s2, dev).unflatten(0, (b2, k2))
        kd = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        kc = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        keys = torch.zeros((b2, k2), dtype=torch.int64, device=dev)
        table = ops.alloc_seen_table(1 << 25, dev)
        fresh = torch.zeros((b2, k2), dtype=torch.uint8, device=dev)
        fused = graph_time(lambda: ops.expand(tgt, tok, out=kids, done=kd, changed=kc, keys=keys), dev, reps=reps)
        kf = keys.clone()
        two = graph_time(lambda: (ops.expand(tgt, tok, out=kids, done=kd, changed=kc),
                                  keys.view(-1).copy_(ops.state_hash(kids.flatten(0, 1)))), dev, reps=reps)
        look = graph_time(lambda: ops.seen(keys, table, mask=kc, fresh=fresh), dev, reps=reps)
        also.append({"workload": f"N2 FUSED tg_expand_keyed_i8: S={s2} batch={b2}, k={k2}: children + their 64-bit keys in one "
                                 f"launch (extend_tree's state_to_str per child, act.py:188-190), then tg_seen_u64 on the keys",
                     "ok": bool(torch.equal(keys, kf)), "us_per_launch": round(fused * 1e6, 2),
                     "us_expand_then_hash": round(two * 1e6, 2), "gain": round(two / fused, 3),
                     "us_seen_lookup": round(look * 1e6, 2), "keys_per_s_lookup": round(b2 * k2 / look, 1)})
        del kids, kd, kc, keys, table, fresh, ring
    return also


SIMDS, SHADER_CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 4  # MI355X: 256 CUs x 4 SIMDs; s_memtime clock; a wave64 VALU op holds a SIMD 4 cycles


def generator_valu_instructions(with_basis):
    """SQ_INSTS_VALU per launch of the fused generator at BASELINE config 5's per-GPU share (S=25, R=64, 4 096 demos)
    from the newest committed rocprofv3 --pmc pass (profiles/rNN_mfma_pmc.json): the instruction count of a launch
    does not depend on the clock, so it can be priced against the time measured live."""
    want = "tg::gen_fused_kernel<25, 2, %s," % ("true" if with_basis else "false")
    for f in sorted((ROOT / "profiles").glob("r*_mfma_pmc.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        for k, e in d.items():
            if k.startswith(want) and "SQ_INSTS_VALU" in e:
                return e["SQ_INSTS_VALU"], f.stem.split("_")[0]
    return None, None


def graph_time(fn, dev, reps, samples=5, warm_ms=60.0):
    """Seconds per call of `fn` (one or more launches on the current stream): `reps` calls captured in a hipGraph,
    replayed back to back, median over `samples` of (HIP events around one replay) / reps.
    Untimed warm-up: the graph is first replayed back to back for `warm_ms` of GPU time.  After an idle gap (the Python
    between two workloads is enough) the MI355X runs a VALU-bound kernel ~20 % slower for the first ~50 ms of sustained
    work (tools/gen_series.py, profiles/r03_generator_series.txt: 34.5 -> 28.3 us per generator launch; the launch- and
    memory-bound step kernels do not move, tools/step_series.py); the figure reported is the sustained one."""
    for _ in range(3):
        fn()
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):  # the launches only: no Python between them
        for _ in range(reps):
            fn()
    cur.wait_stream(side)
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()  # (the first replay uploads the graph: not a measure of its length)
    w0.record()
    g.replay()
    w1.record()
    torch.cuda.synchronize(dev)
    one_ms = max(w0.elapsed_time(w1), 1e-3)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(samples + 1)]
    for _ in range(1 + min(2000, int(warm_ms / one_ms))):  # warm-up and timed replays in ONE stream of back-to-back work
        g.replay()
    for i in range(samples + 1):
        evs[i].record()
        if i < samples:
            g.replay()
    torch.cuda.synchronize(dev)
    return statistics.median(evs[i].elapsed_time(evs[i + 1]) for i in range(samples)) * 1e-3 / reps


if __name__ == "__main__":
    sys.exit(main())
