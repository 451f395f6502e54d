"""ctypes binding of include/tensor_game.h.  The product has NO fallback: if the HIP
library is missing or an entry point is absent, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# TG_LIB_VARIANT=ab selects the A/B build (libtensorgame_ab.so, -DTG_AB_SWITCHES: the same entry points plus the
# TG_* environment switches that force a kernel variant) -- measurement and A/B tests only.
_VARIANT = os.environ.get("TG_LIB_VARIANT", "")  # "" or "ab"
AB_VARIANT = _VARIANT == "ab"
LIB_PATH = Path(__file__).resolve().parent / "lib" / ("libtensorgame_ab.so" if AB_VARIANT else "libtensorgame.so")

TG_ABI_VERSION = 4
TG_MAX_S = 32
TG_MAX_VALUES = 8
TG_MAX_ACTIONS = 4096


class TensorGameError(RuntimeError):
    """A tg_* entry point returned a negative code; the message is tg_last_error()."""

    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed ({code}): {msg}")
        self.code = code


_p, _i, _i64, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64

# name -> argtypes; every symbol include/tensor_game.h declares
SIGNATURES = {
    "tg_abi_version": [],
    "tg_last_error": [],
    "tg_debug_fallbacks": [_p],
    "tg_debug_handovers": [_p],
    "tg_step_i8": [_p, _p, _p, _p, _p, _i64, _i, _i64, _i, _p],
    "tg_step_many_i8": [_p, _p, _p, _p, _p, _i64, _i, _i, _i64, _i, _p],
    "tg_step_stream_i8": [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i64, _i, _p],
    "tg_step_stream_layout": [_i64, _i, _p, _p],
    "tg_step_stream_capacity": [_i, _p],
    "tg_expand_i8": [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i64, _i64, _i, _p],
    "tg_expand_keyed_i8": [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i64, _i64, _i, _p],
    "tg_copy_i8": [_p, _p, _i64, _i, _i64, _i64, _p],
    "tg_done_i8": [_p, _p, _p, _i64, _i, _i64, _p],
    "tg_step_tracked_i8": [_p, _p, _p, _p, _p, _i64, _i, _i64, _i, _p],
    "tg_reset_matmul_i8": [_p, _i64, _i, _i64, _p],
    "tg_reset_broadcast_i8": [_p, _p, _i64, _i, _i64, _p],
    "tg_gen_from_factors_i8": [_p, _p, _p, _i64, _i, _i, _i64, _i, _p],
    "tg_gen_demos_i8": [_p, _p, _p, _i64, _i, _i, _p, _p, _i, _i, _u64, _u64, _p, _i64, _p],
    "tg_sample_basis_i8": [_p, _p, _p, _i64, _i, _p, _p, _i, _u64, _u64, _p],
    "tg_change_basis_i8": [_p, _p, _p, _p, _i64, _i, _i64, _p],
    "tg_emit_frames": [_p, _p, _p, _i, _i64, _i, _i, _i, C.c_float, _i64, _i64, _p],
    "tg_step_emit": [_p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _i, C.c_float, _i64, _i64, _i, _p],
    "tg_hash_u64": [_p, _p, _i64, _i, _i64, _p],
    "tg_seen_u64": [_p, _p, _i64, _p, _p, _p, _i64, _i, _p],
    "tg_rank_i32": [_p, _p, _i64, _i, _i64, _p],
}


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm ships its own libamdhip64 (SONAME libamdhip64.so.7).  Two HIP runtimes in
    one process do not share devices or streams (the second one reports "no ROCm-capable
    device"), so torch's copy must be the one libtensorgame.so binds to: load it first; the
    dynamic linker then satisfies our NEEDED libamdhip64.so.7 by SONAME."""
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so)

    cand = Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def _hip_runtimes_mapped():
    with open("/proc/self/maps") as f:
        return sorted({line.split()[-1] for line in f if "libamdhip64" in line})


def _load() -> C.CDLL:
    _preload_torch_hip_runtime()
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m mat_mul_amd.build [--ab]`). "
            "mat_mul_amd has no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "tg_last_error" else C.c_int
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise ImportError(f"two HIP runtimes are mapped ({rts}); libtensorgame.so must share PyTorch's")
    if lib.tg_abi_version() != TG_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.tg_abi_version()} != {TG_ABI_VERSION}; rebuild it")
    return lib


lib = _load()


def call(name: str, *args) -> None:
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise TensorGameError(name, rc, lib.tg_last_error().decode("utf-8", "replace"))
