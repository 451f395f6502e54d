"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol that
include/tensor_game.h declares, and validates arguments before touching a device."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import mat_mul_amd
from mat_mul_amd import _lib, ops, shard_range

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols(ab=False):
    text = (ROOT / "include" / "tensor_game.h").read_text()
    if not ab:  # drop what the header declares for the A/B build only
        text = re.sub(r"#ifdef TG_AB_SWITCHES.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"^(?:int|const char\*)\s+(tg_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_header_symbols_all_exported():
    syms = declared_symbols()
    assert len(syms) == 25
    assert sorted(_lib.SIGNATURES) == syms                 # the ctypes table covers the header exactly
    lib = C.CDLL(str(_lib.LIB_PATH))
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.tg_abi_version() == 4
    # the A/B variant exports exactly the same entries (it differs only by its environment switches)
    from mat_mul_amd import build
    assert declared_symbols(ab=True) == syms
    ab = C.CDLL(str(build.lib_path(ab=True)))
    for s in syms:
        assert hasattr(ab, s), s


def test_argument_validation_without_gpu():
    lib = _lib.lib
    null = C.c_void_p(0)
    one = C.c_void_p(16)  # never dereferenced: validation fails first
    assert lib.tg_step_i8(one, one, one, one, null, 4, 0, 64, 1, null) == -1          # S out of range
    assert b"S=0" in lib.tg_last_error()
    assert lib.tg_step_i8(one, one, one, one, null, 4, 33, 40000, 1, null) == -1
    assert lib.tg_step_i8(one, one, one, one, null, 4, 4, 63, 1, null) == -1          # stride < S^3
    assert b"stride" in lib.tg_last_error()
    assert lib.tg_step_i8(null, one, one, one, null, 4, 4, 64, 1, null) == -1         # null pointer
    assert lib.tg_step_i8(one, one, one, one, null, -1, 4, 64, 1, null) == -1
    assert lib.tg_step_i8(null, null, null, null, null, 0, 4, 64, 1, null) == 0       # empty batch is a no-op
    assert lib.tg_step_many_i8(one, one, one, one, null, 4, 4, 0, 64, 1, null) == -1  # K = 0
    assert lib.tg_step_many_i8(one, one, one, one, null, 4, 4, 5000, 64, 1, null) == -1
    assert lib.tg_expand_i8(one, one, one, one, null, null, 4, 4, 2, 64, 64, 1, null) == -1  # in == out
    assert lib.tg_reset_matmul_i8(one, 4, 6, 46656, null) == -1                       # n*n > TG_MAX_S
    thr = (C.c_uint32 * 2)(10, 5)
    val = (C.c_int8 * 3)(-1, 0, 1)
    assert lib.tg_gen_demos_i8(one, one, null, 4, 4, 7, thr, val, 3, 1, 0, 0, null, 64, null) == -1  # descending cdf
    thr = (C.c_uint32 * 2)(0, 0xFFFFFFFF)
    val = (C.c_int8 * 3)(-1, 0, 0)
    assert lib.tg_gen_demos_i8(one, one, null, 4, 4, 7, thr, val, 3, 1, 0, 0, null, 64, null) == -1  # never non-zero
    assert lib.tg_change_basis_i8(one, one, one, null, 1, 4, 64, null) == -1           # in-place


def test_ops_refuse_cpu_tensors():
    import torch
    st = torch.zeros((2, 4, 4, 4), dtype=torch.int8)
    ac = torch.ones((2, 12), dtype=torch.int8)
    with pytest.raises(mat_mul_amd.TensorGameError, match="no CPU path"):
        ops.step(st, ac)
    with pytest.raises(mat_mul_amd.TensorGameError):
        mat_mul_amd.TensorGameEnv(4, 4, device="cpu")
    with pytest.raises(mat_mul_amd.TensorGameError, match="int8 range"):
        ops.as_tokens(torch.tensor([[300, 0, 1]]))


def test_thresholds_match_oracle():
    from oracle import tensor_game as O
    for p in [(0.15, 0.7, 0.15), (0.1, 0.8, 0.1), (1, 2, 3, 4), (0.5, 0.5)]:
        assert np.array_equal(ops.categorical_thresholds(p), O.categorical_thresholds(p))
    assert ops.categorical_thresholds((0.15, 0.7, 0.15)).tolist() == [644245094, 3650722202]


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 65536, 2 ** 20 + 3):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """No CPU fallback: without the .so the loader raises, and says how to build it."""
    import importlib
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libtensorgame.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib._load()
    import mat_mul_amd.build as b          # the build module itself never needs the library
    assert b.LIB_PATH.name == "libtensorgame.so"


def test_product_never_imports_oracle():
    for f in (ROOT / "mat_mul_amd").glob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_header_is_plain_c():
    """The drop-in boundary is a C ABI: include/tensor_game.h must compile as C99 (no C++, no torch,
    no HIP headers), so any FFI (cgo, JNI, ctypes, cffi) can consume it."""
    import shutil
    import subprocess
    from pathlib import Path
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    hdr = Path(__file__).resolve().parent.parent / "include" / "tensor_game.h"
    res = subprocess.run([gcc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wpedantic", "-Werror", str(hdr)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_resident_steppers_never_touch_a_register_before_its_load_is_waited_for():
    """tools/audit_pending_loads.py on the cross-compiled assembly: the s4 / s16 stream kernels issue their token loads by inline
    asm with counted waits; between such a load and its wait nothing may read, copy or spill the destination register (the
    compiler does not know the data is still on its way), and the kernels use no scratch.  (hipcc -S: ~30 s.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_pending_loads", ROOT / "tools" / "audit_pending_loads.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    names, problems = mod.audit(mod.assembly())
    assert len(names) == 4 and not problems, problems   # s4 x 2, s4 one-game-per-lane, s16
