"""SURVEY.md section 5 "sanitizers" / VERDICT r3 item 6: the HOST half of libtensorgame.so -- argument validation, dispatch, the
occupancy caches, the sweep counter, the thread-local error string -- under AddressSanitizer + UndefinedBehaviorSanitizer.
`python -m mat_mul_amd.build --hostasan` compiles the three .hip files with `hipcc --cuda-host-only
-fsanitize=address,undefined -fno-sanitize-recover=undefined` (no device code) into libtensorgame_hostasan.so; a child python
with clang's ASan runtime preloaded drives every entry point's error paths through ctypes (no torch, no GPU), from four
threads at once.  GPU AddressSanitizer is not available on this pool: the kernels are covered by the parity tests instead.
Build container only: skipped where hipcc or the ASan runtime is missing."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import ctypes as C, sys, threading
lib = C.CDLL(sys.argv[1])
lib.tg_last_error.restype = C.c_char_p
i64, i32, u64, p, f32 = C.c_int64, C.c_int, C.c_uint64, C.c_void_p, C.c_float
null, one, odd = p(0), p(4096), p(4099)      # never dereferenced: validation fails first, or there is no device
SIG = {
    "tg_step_i8": [p, p, p, p, p, i64, i32, i64, i32, p],
    "tg_step_many_i8": [p, p, p, p, p, i64, i32, i32, i64, i32, p],
    "tg_step_stream_i8": [p, p, p, p, p, p, p, i64, i32, i32, i64, i32, p],
    "tg_step_stream_layout": [i64, i32, p, p],
    "tg_step_stream_capacity": [i32, p],
    "tg_expand_i8": [p, p, p, p, p, p, i64, i32, i32, i64, i64, i32, p],
    "tg_expand_keyed_i8": [p, p, p, p, p, p, p, i64, i32, i32, i64, i64, i32, p],
    "tg_copy_i8": [p, p, i64, i32, i64, i64, p],
    "tg_done_i8": [p, p, p, i64, i32, i64, p],
    "tg_step_tracked_i8": [p, p, p, p, p, i64, i32, i64, i32, p],
    "tg_reset_matmul_i8": [p, i64, i32, i64, p],
    "tg_reset_broadcast_i8": [p, p, i64, i32, i64, p],
    "tg_gen_from_factors_i8": [p, p, p, i64, i32, i32, i64, i32, p],
    "tg_gen_demos_i8": [p, p, p, i64, i32, i32, p, p, i32, i32, u64, u64, p, i64, p],
    "tg_sample_basis_i8": [p, p, p, i64, i32, p, p, i32, u64, u64, p],
    "tg_change_basis_i8": [p, p, p, p, i64, i32, i64, p],
    "tg_emit_frames": [p, p, p, i32, i64, i32, i32, i32, f32, i64, i64, p],
    "tg_step_emit": [p, p, p, p, p, p, i32, i64, i32, i32, i32, f32, i64, i64, i32, p],
    "tg_hash_u64": [p, p, i64, i32, i64, p],
    "tg_seen_u64": [p, p, i64, p, p, p, i64, i32, p],
    "tg_rank_i32": [p, p, i64, i32, i64, p],
    "tg_debug_fallbacks": [p], "tg_debug_handovers": [p],
}
for n, a in SIG.items():
    getattr(lib, n).argtypes = a
    getattr(lib, n).restype = i32
assert lib.tg_abi_version() == 4

def err():
    return lib.tg_last_error().decode()

def validation_round(tag):
    bad = 0
    def neg(rc, want=None):
        nonlocal bad
        assert rc < 0, (tag, rc, err())
        assert err(), tag
        if want is not None:
            assert want in err(), (want, err())
        bad += 1
    # ---- the single step and its relatives
    neg(lib.tg_step_i8(one, one, one, one, null, 4, 0, 64, 1, null), "S=0")
    neg(lib.tg_step_i8(one, one, one, one, null, 4, 33, 40000, 1, null))
    neg(lib.tg_step_i8(one, one, one, one, null, 4, 4, 63, 1, null), "stride")
    neg(lib.tg_step_i8(null, one, one, one, null, 4, 4, 64, 1, null), "null")
    neg(lib.tg_step_i8(one, one, one, one, null, -1, 4, 64, 1, null))
    neg(lib.tg_step_i8(one, one, one, one, null, 1 << 62, 25, 15632, 1, null))          # B * stride overflows nothing
    assert lib.tg_step_i8(null, null, null, null, null, 0, 4, 64, 1, null) == 0
    neg(lib.tg_step_tracked_i8(one, one, odd, one, null, 4, 16, 4096, 1, null), "aligned")
    neg(lib.tg_step_tracked_i8(one, one, null, one, null, 4, 16, 4096, 1, null), "null")
    neg(lib.tg_step_many_i8(one, one, one, one, null, 4, 4, 0, 64, 1, null))
    neg(lib.tg_step_many_i8(one, one, one, one, null, 4, 4, 5000, 64, 1, null))
    neg(lib.tg_expand_i8(one, one, one, one, null, null, 4, 4, 2, 64, 64, 1, null), "in-place")
    neg(lib.tg_expand_i8(one, p(8192), one, one, null, null, 4, 4, 0, 64, 64, 1, null), "k=0")
    neg(lib.tg_expand_keyed_i8(one, p(8192), one, one, null, null, odd, 4, 4, 2, 64, 64, 1, null), "8-byte")
    neg(lib.tg_copy_i8(one, one, 4, 4, 63, 64, null))
    neg(lib.tg_done_i8(null, one, null, 4, 4, 64, null))
    # ---- the streamed stepper: layout, capacity, argument checks (the occupancy queries run without a device)
    units, gpu, cap = i64(0), i32(0), i64(0)
    neg(lib.tg_step_stream_layout(-1, 4, C.byref(units), C.byref(gpu)))
    neg(lib.tg_step_stream_layout(16, 9, C.byref(units), C.byref(gpu)), "S=9")
    assert lib.tg_step_stream_layout(1000, 16, C.byref(units), C.byref(gpu)) == 0 and units.value == 1000 and gpu.value == 1
    assert lib.tg_step_stream_layout(100, 4, C.byref(units), C.byref(gpu)) == 0 and units.value * gpu.value >= 100
    assert lib.tg_step_stream_layout(100, 4, None, None) == 0
    neg(lib.tg_step_stream_layout(1 << 40, 4, C.byref(units), C.byref(gpu)), "resident")
    neg(lib.tg_step_stream_capacity(9, C.byref(cap)))
    neg(lib.tg_step_stream_capacity(4, None), "null")
    for S in (4, 16, 25):
        assert lib.tg_step_stream_capacity(S, C.byref(cap)) == 0 and cap.value > 0
    neg(lib.tg_step_stream_i8(one, one, one, null, null, null, null, 4, 4, 0, 64, 1, null), "K=0")
    neg(lib.tg_step_stream_i8(one, one, one, null, null, null, null, 4, 9, 3, 736, 1, null), "S=9")
    neg(lib.tg_step_stream_i8(one, one, one, null, odd, null, null, 4, 4, 3, 64, 1, null), "4-byte")
    neg(lib.tg_step_stream_i8(odd, one, one, null, null, null, null, 4, 4, 3, 64, 1, null), "aligned")
    neg(lib.tg_step_stream_i8(one, one, one, null, one, null, null, 1 << 40, 16, 3, 4096, 1, null))
    neg(lib.tg_step_stream_i8(one, one, one, null, one, null, null, cap.value + 1, 25, 3, 15632, 1, null), "resident")
    assert lib.tg_step_stream_i8(null, null, null, null, null, null, null, 0, 4, 3, 64, 1, null) == 0
    # ---- resets, generator, basis
    neg(lib.tg_reset_matmul_i8(one, 4, 6, 46656, null))
    neg(lib.tg_reset_matmul_i8(null, 4, 2, 64, null))
    neg(lib.tg_reset_broadcast_i8(one, null, 4, 4, 64, null))
    thr = (C.c_uint32 * 2)(10, 5)
    val = (C.c_int8 * 3)(-1, 0, 1)
    neg(lib.tg_gen_demos_i8(one, one, null, 4, 4, 7, thr, val, 3, 1, 0, 0, null, 64, null))       # descending cdf
    thr2 = (C.c_uint32 * 2)(0, 0xFFFFFFFF)
    val2 = (C.c_int8 * 3)(-1, 0, 0)
    neg(lib.tg_gen_demos_i8(one, one, null, 4, 4, 7, thr2, val2, 3, 1, 0, 0, null, 64, null))     # never non-zero
    neg(lib.tg_gen_demos_i8(one, one, null, 4, 4, 7, thr, val, 9, 1, 0, 0, null, 64, null))       # too many values
    neg(lib.tg_gen_demos_i8(one, one, null, 4, 4, 0, thr, val, 3, 1, 0, 0, null, 64, null))       # R = 0
    neg(lib.tg_gen_from_factors_i8(null, one, null, 4, 4, 3, 64, 1, null))
    neg(lib.tg_sample_basis_i8(null, null, null, 4, 4, thr, val, 3, 0, 0, null))
    neg(lib.tg_change_basis_i8(one, one, one, null, 1, 4, 64, null), "in-place")
    # ---- model input, keys, rank
    neg(lib.tg_emit_frames(one, one, null, 7, 4, 4, 1, 0, 1.0, 64, 64, null))                     # unknown dtype
    neg(lib.tg_emit_frames(one, one, null, 0, 4, 4, 0, 0, 1.0, 64, 64, null))                     # T = 0
    neg(lib.tg_emit_frames(one, one, null, 0, 4, 4, 2, 5, 1.0, 64, 128, null))                    # head slot >= T
    neg(lib.tg_step_emit(one, one, one, null, null, null, 0, 4, 4, 2, 9, 1.0, 64, 128, 1, null))
    neg(lib.tg_hash_u64(one, null, 4, 4, 64, null), "null")
    neg(lib.tg_seen_u64(odd, one, 8, null, null, null, 4, 0, null))
    neg(lib.tg_seen_u64(one, one, 7, null, null, null, 4, 1, null))                               # capacity not a power of two
    neg(lib.tg_rank_i32(one, null, 4, 4, 64, null))
    neg(lib.tg_debug_fallbacks(None))
    neg(lib.tg_debug_handovers(None))
    # ---- valid arguments reach the launch: without a device that is a HIP error code and a message, never a crash
    for rc in (lib.tg_step_i8(one, one, one, one, null, 4, 4, 64, 1, null),
               lib.tg_step_i8(one, one, one, one, null, 4, 16, 4096, 1, null),
               lib.tg_step_i8(one, one, one, one, null, 4, 25, 15632, 1, null),
               lib.tg_expand_i8(one, p(1 << 20), one, one, null, null, 4, 16, 8, 4096, 4096, 1, null),
               lib.tg_copy_i8(one, p(1 << 20), 4, 4, 64, 64, null),
               lib.tg_done_i8(one, one, null, 4, 9, 736, null),
               lib.tg_hash_u64(one, one, 4, 9, 736, null),
               lib.tg_reset_matmul_i8(one, 4, 2, 64, null),
               lib.tg_gen_demos_i8(one, one, null, 4, 25, 64, (C.c_uint32 * 2)(644245094, 3650722202), val, 3, 1, 5, 0, null, 15632, null),
               lib.tg_step_stream_i8(one, one, one, null, null, null, null, 64, 4, 3, 64, 1, null)):
        assert rc <= 0
        if rc < 0:
            assert err()
    return bad

counts = []
def worker(t):
    for _ in range(3):
        counts.append(validation_round(t))
ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
for t in ths: t.start()
for t in ths: t.join()
assert len(counts) == 12 and len(set(counts)) == 1 and counts[0] >= 40, counts
print("HOSTASAN_OK", counts[0])
'''


def test_host_half_of_the_library_under_asan_and_ubsan(tmp_path):
    from mat_mul_amd import build

    try:
        rt = build.asan_runtime()
    except RuntimeError:
        pytest.skip("hipcc not available")
    if rt is None:
        pytest.skip("clang's shared AddressSanitizer runtime is not installed")
    lib = build.build(ab="hostasan")
    script = tmp_path / "hostasan_child.py"
    script.write_text(CHILD)
    env = dict(os.environ, LD_PRELOAD=str(rt), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("TG_LIB_VARIANT", None)
    res = subprocess.run([sys.executable, str(script), str(lib)], env=env, capture_output=True, text=True, timeout=600)
    report = res.stdout[-3000:] + res.stderr[-6000:]
    assert res.returncode == 0 and "HOSTASAN_OK" in res.stdout, report
    assert "ERROR: AddressSanitizer" not in report and "runtime error:" not in report, report
