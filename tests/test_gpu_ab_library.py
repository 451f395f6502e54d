"""The A/B library (libtensorgame_ab.so, -DTG_AB_SWITCHES) on the GPU: the measurement switches force kernel variants
the product dispatch takes only at other sizes (or no longer takes); whatever they compute must stay bit-exact too.  Every case runs in a child process with TG_LIB_VARIANT=ab, because a process
binds ONE variant of the library when mat_mul_amd is first imported."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

AB_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mat_mul_amd import ops, _lib
from oracle import tensor_game as O
assert _lib.AB_VARIANT and "libtensorgame_ab.so" in open("/proc/self/maps").read()
rng = np.random.default_rng(1)
for S, B, K in [(9, 9, 6), (16, 6, 5), (25, 3, 7)]:
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, K, 3 * S)).astype(np.int8)
    t = ops.alloc_states(B, S, "cuda:0"); t.copy_(torch.from_numpy(st))
    a = torch.from_numpy(ac).cuda()
    w, wd, _ = O.step_i8(st, ac[:, 0])
    o, d = ops.step(t, a[:, 0].contiguous())
    assert np.array_equal(o.cpu().numpy(), w) and np.array_equal(d.cpu().numpy(), wd)
    w, wds, _ = O.step_many_i8(st, ac)
    o, ds = ops.step_many(t, a)
    assert np.array_equal(o.cpu().numpy(), w) and np.array_equal(ds.cpu().numpy(), wds)
    wk, wdn, wch, _ = O.expand_i8(st, ac)
    k, dn, ch = ops.expand(t, a)
    assert np.array_equal(k.cpu().numpy(), wk) and np.array_equal(dn.cpu().numpy(), wdn) and np.array_equal(ch.cpu().numpy(), wch)
    assert np.array_equal(ops.gen_from_factors(a, S).cpu().numpy(), O.gen_from_factors_i8(ac)[0])
    thr = O.categorical_thresholds((0.15, 0.7, 0.15))
    P_o, _, _ = O.sample_basis(B, S, O.categorical_thresholds((0.05, 0.9, 0.05)), (-1, 0, 1), seed=4)
    tok_o, tgt_o, ovf_o = O.gen_demos_i8(B, S, 40, thr, (-1, 0, 1), 1, seed=3, basis=P_o)
    ovf = torch.zeros(B, dtype=torch.uint8, device="cuda:0")
    tok, tgt = ops.gen_demos(B, S, 40, "cuda:0", seed=3, basis=torch.from_numpy(P_o.astype(np.int8)).cuda(), overflow=ovf)
    assert np.array_equal(tok.cpu().numpy(), tok_o) and np.array_equal(tgt.cpu().numpy(), tgt_o)
    assert np.array_equal(ovf.cpu().numpy(), ovf_o)
    for shift in (1, 2):   # the plain ternary generator: byte products by table lookup unless TG_GF_NO_LUT
        tok_o, tgt_o, _ = O.gen_demos_i8(B, S, 40, thr, (-1, 0, 1), shift, seed=5)
        tok, tgt = ops.gen_demos(B, S, 40, "cuda:0", seed=5, shift=shift)
        assert np.array_equal(tok.cpu().numpy(), tok_o) and np.array_equal(tgt.cpu().numpy(), tgt_o), (S, shift)
print("AB_OK")
'''

STEP_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mat_mul_amd import ops, _lib
from oracle import tensor_game as O
assert _lib.AB_VARIANT
DEV = "cuda:0"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
host = lambda t: t.detach().cpu().numpy()
def padded(st):
    t = ops.alloc_states(st.shape[0], st.shape[1], DEV); t.copy_(torch.from_numpy(np.ascontiguousarray(st))); return t
for S, B in [(16, 1), (16, 3), (16, 131), (25, 2), (25, 37), (9, 1), (9, 70), (4, 1), (4, 3), (4, 70), (4, 257)]:
    rng = np.random.default_rng(S * 7 + B)
    for case in ("sparse", "dense", "wide", "overflow", "null"):
        st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
        ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
        if case == "dense":
            ac = rng.integers(0, 3, size=(B, 3 * S)).astype(np.int8)          # > 64 candidate rows per game
        if case == "wide":
            ac[::2] = rng.integers(-128, 128, size=ac[::2].shape)                # beyond the 16-bit form
        if case == "overflow":
            st = rng.choice([-128, 127, 0, 1], size=st.shape).astype(np.int8)
        if case == "null":
            ac[:, :S] = 1                                                        # u == 0: nothing changes
            st[::2] = 0                                                          # and these are already done
        for shift in ((1, -2, 127) if case == "wide" else (1,)):
            want, want_done, want_ovf = O.step_i8(st, ac, shift=shift)
            for inplace in (False, True):
                t = padded(st)
                ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
                out, done = ops.step(t, dev(ac), out=t if inplace else None, overflow=ovf, shift=shift)
                assert np.array_equal(host(out), want), (S, B, case, shift, inplace)
                assert np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf), (S, B, case, shift, inplace)
                assert inplace or np.array_equal(host(t), st)
# the tracked step: both of its S = 25 kernels at a small batch (TG_TRACKED_SPARSE / TG_TRACKED_FULL pick one)
for S, B in [(25, 9), (16, 5)]:
    rng = np.random.default_rng(S + B)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    t = padded(st)
    _, nnz = ops.done(t, want_nnz=True)
    cur = st
    for k in range(3):
        ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
        if k == 1:
            ac[::2] = rng.integers(-100, 100, size=ac[::2].shape)
        cur, want_done, _ = O.step_i8(cur, ac)
        _, done = ops.step_tracked(t, dev(ac), nnz)
        assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done), (S, B, k)
        assert np.array_equal(host(nnz), np.count_nonzero(cur.reshape(B, -1), axis=1)), (S, B, k)
print("STEP_OK")
'''


def _run(script_text, tmp_path, marker, extra_env=None):
    script = tmp_path / "ab_case.py"
    script.write_text(script_text)
    env = dict(os.environ, TG_LIB_VARIANT="ab", **(extra_env or {}))
    res = subprocess.run([sys.executable, str(script), str(ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and marker in res.stdout, (res.stdout[-1000:], res.stderr[-3000:])


@pytest.mark.parametrize("env_name", ["TG_NO_ROWS", "TG_NO_S16_DIRECT", "TG_NO_MFMA", "TG_MFMA_MANY_ALWAYS", "TG_NO_FUSED_GEN", "TG_GF_NO_LUT"])
def test_ab_switch_paths_stay_exact(env_name, tmp_path):
    """The measurement switches (packed chunks instead of rows; vector ALU instead of the matrix cores) select
    kernels that the product dispatch no longer uses at these shapes -- they must stay bit-exact."""
    _run(AB_SCRIPT, tmp_path, "AB_OK", {env_name: "1"})


@pytest.mark.parametrize("env_names", ["TG_S16_LINES", "TG_S16_NT_LOADS", "TG_S25_LINES", "TG_S25_NT_LOADS", "TG_NO_S25_DIRECT",
                                       "TG_NO_S16_DIRECT", "TG_NO_S9_DIRECT", "TG_S4_NT_LOADS", "TG_S4_NT_LOADS TG_S4_TOKEN_WAIT",
                                       "TG_TRACKED_SPARSE", "TG_TRACKED_FULL", "TG_S16_NO_DIGITS", "TG_S16_LINES TG_S16_NO_DIGITS", "TG_S16_NT_LOADS TG_S16_NO_DIGITS",
                                       "TG_S4_NO_DIGITS", "TG_S4_NT_LOADS TG_S4_NO_DIGITS", "TG_S4_NT_LOADS TG_S4_TOKEN_WAIT TG_S4_NO_DIGITS"])
def test_single_step_variants_stay_exact(env_names, tmp_path):
    """The step variants the product takes by footprint only, forced here at small batches: S=16 / S=25 with whole-line
    stores (from 96 MiB of states on) and non-temporal state loads on top (320 MiB .. 1.5 GiB), S=4 with non-temporal
    loads (from 96 MiB on) and with the token awaited first (from 1 GiB on), S=4 in the packed int16 form alone (the
    product tries the digit form first), and the staged kernels that the direct
    S=9 / S=16 / S=25 step kernels replaced: sparse, dense (more candidate rows than the queue holds), wide-factor,
    overflowing and null actions, in place and out of place."""
    _run(STEP_SCRIPT, tmp_path, "STEP_OK", {n: "1" for n in env_names.split()})


STREAM_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mat_mul_amd import ops, _lib
from oracle import tensor_game as O
assert _lib.AB_VARIANT
DEV = "cuda:0"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
host = lambda t: t.detach().cpu().numpy()
def padded(st):
    t = ops.alloc_states(st.shape[0], st.shape[1], DEV); t.copy_(torch.from_numpy(np.ascontiguousarray(st))); return t
rng = np.random.default_rng(23)
for S, B, k in [(4, 70, 8), (4, 5, 3), (16, 9, 4), (9, 19, 5), (25, 3, 3)]:
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, k, 3 * S)).astype(np.int8)
    ac[::3, 0] = rng.integers(-40, 40, size=ac[::3, 0].shape)                       # wide factors -> 32-bit redo
    kids_o, done_o, chg_o, ovf_o = O.expand_i8(st, ac)
    ovf = torch.zeros((B, k), dtype=torch.uint8, device=DEV)
    kids, done, chg, keys = ops.expand(padded(st), dev(ac), overflow=ovf, want_keys=True)
    assert np.array_equal(host(kids), kids_o) and np.array_equal(host(done), done_o), (S, B, k)
    assert np.array_equal(host(chg), chg_o) and np.array_equal(host(ovf), ovf_o), (S, B, k)
    assert np.array_equal(host(keys).view(np.uint64), O.state_hash(kids_o.reshape(B * k, S, S, S)).reshape(B, k))
    # copy: same and different strides, guard bytes behind every game untouched
    src = padded(st)
    wide = torch.full((B, S ** 3 + 48), 77, dtype=torch.int8, device=DEV)
    dst = wide[:, :S ** 3].unflatten(1, (S, S, S))
    ops.copy_states(src, dst)
    assert np.array_equal(host(dst), st) and bool((wide[:, S ** 3:] == 77).all())
    assert np.array_equal(host(ops.copy_states(src)), st)
    # model-input frames
    T = 3
    ring = ops.alloc_ring(B, S, T, DEV)
    frames = rng.integers(-128, 128, size=(B, T, S, S, S)).astype(np.int8)
    ring.copy_(dev(frames))
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x, sc = ops.emit_frames(ring, 1, 4.0, dt)
        want = frames[:, [1, 0, 2]].astype(np.float32)                               # newest first from head slot 1
        assert np.array_equal(x.float().cpu().numpy(), want) and bool((sc == 4.0).all()), (S, dt)
    # fused step + model input (one kernel at S=4; its non-temporal form is reached through TG_EMIT_NT only)
    ac1 = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, 3 * S)).astype(np.int8)
    ring.copy_(dev(frames))
    x, sc, dn, nxt = ops.step_emit(ring, 1, dev(ac1), 2.0, dtype=torch.float16)
    new, want_done, _ = O.step_i8(frames[:, 1], ac1)
    want = frames.copy(); want[:, 2] = new
    assert nxt == 2 and np.array_equal(host(ring), want) and np.array_equal(host(dn), want_done), S
    assert np.array_equal(x.float().cpu().numpy(), want[:, [2, 1, 0]].astype(np.float32)), S
print("STREAM_OK")
'''


@pytest.mark.parametrize("env_name", ["TG_EXPAND_NT", "TG_EXPAND_NO_NT", "TG_COPY_NT1", "TG_COPY_NT2", "TG_COPY_PLAIN", "TG_EMIT_NT", "TG_STEP_EMIT_FUSED"])
def test_write_stream_variants_stay_exact(env_name, tmp_path):
    """The non-temporal forms of the write streams -- expand's children (from 128 MiB of children on), the copy's loads and
    stores (by footprint) and the model-input frames (from 128 MiB of output on) -- forced at small batches against the
    oracle; the product reaches them only through full-size property tests."""
    _run(STREAM_SCRIPT, tmp_path, "STREAM_OK", {env_name: "1"})


LANES_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mat_mul_amd import ops, _lib
from oracle import tensor_game as O
assert _lib.AB_VARIANT
DEV = "cuda:0"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
host = lambda t: t.detach().cpu().numpy()
S = 4
for B, K in [(1, 3), (63, 5), (64, 9), (65, 17), (1000, 14), (4099, 25)]:
    rng = np.random.default_rng(B * 31 + K)
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
    st[::5] = O.action_to_tensor(ac[0, ::5]).astype(np.int8)
    if B > 3:
        st[3] = 127
        ac[1, 3] = 0                                                       # overflows at step 1
        ac[2, 2] = rng.integers(-3, 6, size=3 * S)                         # wide factors: the general form
        ac[K - 1, 1::4] = rng.integers(0, 3, size=ac[K - 1, 1::4].shape)
        ac[:, B - 1] = rng.integers(-128, 128, size=(K, 3 * S))            # full-range tokens in the last (ragged) unit
    for shift in (1, 2, -1):
        want_done, want_ovf, cur = np.zeros((K, B), np.uint8), np.zeros(B, np.uint8), st.copy()
        for k in range(K):
            cur, d, o = O.step_i8(cur, ac[k], shift=shift)
            want_done[k] = d
            want_ovf |= o
        want_lanes = "TG_STREAM_NO_LANES" not in __import__("os").environ
        n_units, gpu = ops.step_stream_layout(B, S, DEV)
        assert (gpu == 64) == want_lanes and n_units == -(-B // gpu)
        for ready, pad in ((None, 16), (torch.ones(K, dtype=torch.int32, device=DEV), 16), (None, 48)):   # (48: a 96-byte game stride)
            t = ops.alloc_states(B, S, DEV, pad_to=pad); t.copy_(dev(st))
            assert B == 1 or t.stride(0) == (64 if pad == 16 else 96)
            ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
            prog = torch.zeros(n_units, dtype=torch.int32, device=DEV)
            status = torch.zeros(1, dtype=torch.int32, device=DEV)
            out, done = ops.step_stream(t, dev(ac), overflow=ovf, ready=ready, progress=prog, status=status, shift=shift)
            torch.cuda.synchronize()
            assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done) and np.array_equal(host(ovf), want_ovf), (B, K, shift)
            assert bool((prog == K).all()) and int(status[0]) == 0
# released step by step from a second stream (blocks of one: the serial order), then in bursts
B, K = 300, 21
rng = np.random.default_rng(5)
st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(K, B, 3 * S)).astype(np.int8)
want_done, cur = np.zeros((K, B), np.uint8), st.copy()
for k in range(K):
    cur, want_done[k], _ = O.step_i8(cur, ac[k])
for bursts in ((1,) * K, (1, 4, 2, 9, 1, 3, 8, 8)):
    t = ops.alloc_states(B, S, DEV); t.copy_(dev(st))
    ready = torch.zeros(K, dtype=torch.int32, device=DEV)
    ready[:3] = 1; ready[4:6] = 1; ready[K - 1] = 1
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    prog = torch.zeros(ops.step_stream_layout(B, S, DEV)[0], dtype=torch.int32, device=DEV)
    done = torch.zeros((K, B), dtype=torch.uint8, device=DEV)
    acd = dev(ac)
    torch.cuda.synchronize()
    side, prod = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(side):
        ops.step_stream(t, acd, done=done, ready=ready, progress=prog, status=status)
    with torch.cuda.stream(prod):
        k = 3
        for b in bursts:
            ready[k:min(K, k + b)].fill_(1)
            prod.synchronize()
            k += b
            if k >= K:
                break
    side.synchronize()
    assert int(status[0]) == 0
    assert np.array_equal(host(t), cur) and np.array_equal(host(done), want_done) and bool((prog == K).all())
# never released beyond step 1: every wavefront gives up after its time bound, the state stops where the words stopped
t = ops.alloc_states(B, S, DEV); t.copy_(dev(st))
ready = torch.zeros(K, dtype=torch.int32, device=DEV); ready[:2] = 1
status = torch.zeros(1, dtype=torch.int32, device=DEV)
prog = torch.zeros(ops.step_stream_layout(B, S, DEV)[0], dtype=torch.int32, device=DEV)
import time
t0 = time.perf_counter()
ops.step_stream(t, dev(ac), ready=ready, progress=prog, status=status)
torch.cuda.synchronize()
el = time.perf_counter() - t0
two = st.copy()
for k in range(2):
    two, _, _ = O.step_i8(two, ac[k])
assert int(status[0]) == 1 and np.array_equal(host(t), two) and bool((prog == 2).all()) and 0.9 < el < 3.0, (int(status[0]), el)
print("LANES_OK")
'''


@pytest.mark.parametrize("env_name", ["TG_STREAM_LANES", "TG_STREAM_NO_LANES"])
def test_stream_lane_kernel_forced_at_small_batches(env_name, tmp_path):
    """Round 4: s4_stream_kernel_lanes (one game per lane, tokens double-buffered, counted waits over an exact number of
    stores) serves tg_step_stream_i8 from 57 344 games on; TG_STREAM_LANES forces it at every batch -- single games, ragged
    units, shifts 1 / 2 / -1 (the last two leave the digit form: the whole wavefront takes the general form on its LDS
    image), overflow, full-range tokens, ready words pre-set, released one by one and in bursts.  TG_STREAM_NO_LANES: the
    four-lanes-per-game kernels on the same cases."""
    _run(LANES_SCRIPT, tmp_path, "LANES_OK", {env_name: "1"})


def test_product_library_has_no_switches():
    """The product library reads no environment variable and does not export the A/B-only entry."""
    import ctypes as C

    from mat_mul_amd import _lib, build

    assert not _lib.AB_VARIANT
    lib = C.CDLL(str(build.LIB_PATH))
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(build.LIB_PATH)], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout
