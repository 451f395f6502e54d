"""Build libtensorgame.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libtensorgame.so"
SOURCES = [CSRC / "tg_kernels.hip", CSRC / "tg_gen.hip", CSRC / "tg_aux.hip"]
HEADERS = [CSRC / "tg_device.h", CSRC / "tg_packed.h", CSRC / "tg_rows.h", CSRC / "tg_mfma.h", PKG.parent / "include" / "tensor_game.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source into mat_mul_amd/lib/libtensorgame.so (gfx950 only)."""
    if not force and not is_stale():
        return LIB_PATH
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    cmd = [_hipcc(), "-O3", "-std=c++17", "-shared", "-fPIC", "--offload-arch=gfx950",
           "-mllvm", "-amdgpu-mfma-vgpr-form",  # MFMA results in VGPRs (tg_mfma.h): no v_accvgpr moves
           "-Wall", "-Wno-unused-function", *map(str, SOURCES), "-o", str(LIB_PATH)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
