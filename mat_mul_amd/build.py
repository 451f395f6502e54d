"""Build libtensorgame.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

Every HIP source is compiled to its own object (in parallel, re-compiled only when it or a header
is newer) and the objects are linked into ``mat_mul_amd/lib/libtensorgame.so``.

Two variants:
  * the PRODUCT library (default): no environment switches, no measurement-only kernels;
  * the A/B library ``libtensorgame_ab.so`` (``build(ab=True)``, compiled with -DTG_AB_SWITCHES): the
    same sources and entry points plus the getenv switches that force a kernel variant.  Loaded only when
    TG_LIB_VARIANT=ab is set (tests/test_gpu_ab_library.py, tools/).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
OBJ_DIR = LIB_DIR / "obj"
LIB_PATH = LIB_DIR / "libtensorgame.so"
LIB_PATH_AB = LIB_DIR / "libtensorgame_ab.so"
# The HOST half of the library (validation, dispatch, occupancy caches, error strings: everything outside the kernels) under
# AddressSanitizer + UndefinedBehaviorSanitizer: `-fsanitize=address,undefined -fno-gpu-sanitize` instruments the host pass
# only (the device pass compiles as usual: the host objects need its code object to link; GPU ASan is not available on
# this pool).  CPU build container only (tests/test_host_sanitizer_cpu.py); never loaded by the package, never shipped.
LIB_PATH_HOSTASAN = LIB_DIR / "libtensorgame_hostasan.so"
SAN_FLAGS = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form",
             "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
             "-Wall", "-Wno-unused-function"]
SOURCES = sorted(CSRC.glob("*.hip"))
HEADERS = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "tensor_game.h"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
         "-mllvm", "-amdgpu-mfma-vgpr-form",  # MFMA results in VGPRs (tg_mfma.h): no v_accvgpr moves
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def lib_path(ab=False) -> Path:
    if ab == "hostasan":
        return LIB_PATH_HOSTASAN
    return LIB_PATH_AB if ab else LIB_PATH


def asan_runtime() -> Path | None:
    """clang's shared AddressSanitizer runtime next to hipcc's clang (to LD_PRELOAD into a python that loads the
    sanitized library), or None."""
    res = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = Path(res.stdout.strip()) if res.returncode == 0 else None
    if p and p.is_absolute() and p.exists():
        return p
    for cand in sorted(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.so")):
        return cand
    return None


def _newest_input() -> float:
    return max(p.stat().st_mtime for p in SOURCES + HEADERS + [Path(__file__)])


def is_stale(ab=False) -> bool:
    lib = lib_path(ab)
    return (not lib.exists()) or lib.stat().st_mtime < _newest_input()


def _obj(src: Path, ab) -> Path:
    return OBJ_DIR / f"{src.stem}{'.hostasan' if ab == 'hostasan' else '.ab' if ab else ''}.o"


def build(force: bool = False, verbose: bool = False, ab=False) -> Path:
    """Compile every HIP source (gfx950 only) and link the chosen variant of the library."""
    lib = lib_path(ab)
    if not force and not is_stale(ab):
        return lib
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    hdr_time = max(p.stat().st_mtime for p in HEADERS + [Path(__file__)])
    extra = [] if ab == "hostasan" or not ab else ["-DTG_AB_SWITCHES"]

    def compile_one(src: Path):
        obj = _obj(src, ab)
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_time):
            return src, 0.0, None
        cmd = [hipcc, *(SAN_FLAGS if ab == "hostasan" else FLAGS), *extra, "-c", str(src), "-o", str(obj)]
        t0 = time.perf_counter()
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return src, time.perf_counter() - t0, f"{' '.join(cmd)}\n{res.stdout}\n{res.stderr}"
        if verbose and res.stderr.strip():
            print(res.stderr.strip())
        return src, time.perf_counter() - t0, None

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    for src, dt, err in results:
        if err:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{err}")
        if verbose:
            print(f"  {src.name}: {'up to date' if dt == 0.0 else f'{dt:.1f} s'}")
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *[str(_obj(s, ab)) for s in SOURCES], "-o", str(lib)]
    if ab == "hostasan":
        cmd += ["-fsanitize=address,undefined", "-shared-libasan"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return lib


if __name__ == "__main__":
    import sys

    print(build(force="--force" in sys.argv, verbose=True,
                ab="hostasan" if "--hostasan" in sys.argv else "--ab" in sys.argv))
