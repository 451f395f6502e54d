"""The A/B library (libtensorgame_ab.so, -DTG_AB_SWITCHES) on the GPU: the measurement switches, the
first-generation 32-bit kernels and tg_step_sparse_i8 are not part of the product library, but whatever they
compute must stay bit-exact too.  Every case runs in a child process with TG_LIB_VARIANT=ab, because a process
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
print("AB_OK")
'''

SPARSE_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mat_mul_amd import TensorGameEnv, ops, _lib
from oracle import tensor_game as O
assert _lib.AB_VARIANT
DEV = "cuda:0"
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
host = lambda t: t.detach().cpu().numpy()
def padded(st):
    t = ops.alloc_states(st.shape[0], st.shape[1], DEV); t.copy_(torch.from_numpy(np.ascontiguousarray(st))); return t
for S, B in [(4, 70), (9, 33), (16, 12), (25, 5), (6, 9), (16, 1)]:
    rng = np.random.default_rng(S * 11 + B)
    K = 9
    ac = rng.choice([0, 1, 2], p=[0.15, 0.7, 0.15], size=(B, K, 3 * S)).astype(np.int8)
    ac[:, 3] = rng.integers(0, 3, size=(B, 3 * S))                       # a dense action
    ac[1::3, 5] = rng.integers(-3, 6, size=ac[1::3, 5].shape)            # wide factors
    st = rng.integers(-2, 3, size=(B, S, S, S)).astype(np.int8)
    st[::4] = O.gen_from_factors_i8(ac[::4, :4])[0]                      # these reach zero after step 3
    st[2::5] = rng.choice([-128, 127, 0], size=st[2::5].shape)           # these overflow
    for layout in ("padded", "packed"):
        t = padded(st) if layout == "padded" else dev(st)
        nnz = ops.done(t, want_nnz=True)[1]
        ovf = torch.zeros(B, dtype=torch.uint8, device=DEV)
        cur, want_ovf = st.copy(), np.zeros(B, np.uint8)
        for k in range(K):
            cur, want_done, o = O.step_i8(cur, ac[:, k])
            want_ovf |= o
            out, done = ops.step_sparse(t, dev(ac[:, k]), nnz, overflow=ovf)
            assert out.data_ptr() == t.data_ptr()
            assert np.array_equal(host(t), cur), (S, layout, k)
            assert np.array_equal(host(done), want_done), (S, layout, k)
            assert np.array_equal(host(nnz), O.nnz_per_game(cur)), (S, layout, k)
            assert np.array_equal(host(ovf), want_ovf), (S, layout, k)
        assert want_ovf.any() or B == 1
    env = TensorGameEnv(B, S, DEV, incremental=True)
    env.reset(dev(st))
    cur = st.copy()
    for k in range(4):
        cur, want_done, _ = O.step_i8(cur, ac[:, k])
        state, done = env.step(dev(ac[:, k]))
        assert np.array_equal(host(state), cur) and np.array_equal(host(done), want_done)
        assert np.array_equal(host(env.nnz()), O.nnz_per_game(cur))
    idx = [b for b in range(0, B, 4) if b % 5 != 2]                       # terminating games not overwritten above
    assert host(env.done)[idx].all()
print("SPARSE_OK")
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
for S, B in [(16, 1), (16, 3), (16, 131), (25, 2), (25, 37), (9, 1), (9, 70)]:
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
print("STEP_OK")
'''


def _run(script_text, tmp_path, marker, extra_env=None):
    script = tmp_path / "ab_case.py"
    script.write_text(script_text)
    env = dict(os.environ, TG_LIB_VARIANT="ab", **(extra_env or {}))
    res = subprocess.run([sys.executable, str(script), str(ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and marker in res.stdout, (res.stdout[-1000:], res.stderr[-3000:])


@pytest.mark.parametrize("env_name", ["TG_FORCE_I32", "TG_NO_ROWS", "TG_NO_S16_DIRECT", "TG_NO_MFMA", "TG_MFMA_MANY_ALWAYS", "TG_NO_FUSED_GEN"])
def test_ab_switch_paths_stay_exact(env_name, tmp_path):
    """The measurement switches (32-bit cursor kernels; packed chunks instead of rows; vector ALU instead of
    the matrix cores) select kernels that the product dispatch no longer uses -- they must stay bit-exact."""
    _run(AB_SCRIPT, tmp_path, "AB_OK", {env_name: "1"})


@pytest.mark.parametrize("env_name", ["TG_S16_LINES", "TG_S16_NT_LOADS", "TG_S25_LINES", "TG_S25_NT_LOADS", "TG_NO_S25_DIRECT", "TG_NO_S16_DIRECT",
                                      "TG_NO_S9_DIRECT"])
def test_single_step_variants_stay_exact(env_name, tmp_path):
    """The S=16 / S=25 steps with whole-line stores (the product takes them from 96 MiB of states on; forced here at small
    batches) and with non-temporal state loads on top (from 320 MiB on), and the staged kernels that the direct S=9 / S=16 / S=25 step kernels replaced: sparse, dense (more candidate
    rows than the queue holds), wide-factor, overflowing and null actions, in place and out of place."""
    _run(STEP_SCRIPT, tmp_path, "STEP_OK", {env_name: "1"})


def test_step_sparse_rollout_matches_dense(tmp_path):
    """tg_step_sparse_i8 (A/B library only) over multi-step rollouts: state, done and the carried nnz equal the
    oracle at every step -- sparse, dense and wide-factor actions, games that terminate, games that overflow."""
    _run(SPARSE_SCRIPT, tmp_path, "SPARSE_OK")


def test_product_library_has_no_switches():
    """The product library reads no environment variable and does not export the A/B-only entry."""
    import ctypes as C

    from mat_mul_amd import _lib, build

    assert not _lib.AB_VARIANT
    lib = C.CDLL(str(build.LIB_PATH))
    assert not hasattr(lib, "tg_step_sparse_i8")
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(build.LIB_PATH)], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout
