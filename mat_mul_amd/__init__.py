"""mat_mul_amd -- MI355X-native batched tensor-decomposition environment.

The hot path of kurtosis/mat_mul (AlphaTensor re-implementation) -- the per-game state update
``state <- state - u(x)v(x)w`` with the all-zero terminal check, and the synthetic-demonstration
generator -- as hand-written HIP kernels for gfx950 behind the C ABI of ``include/tensor_game.h``.

There is no CPU fallback: the first use of anything below loads
``mat_mul_amd/lib/libtensorgame.so`` (through ``mat_mul_amd._lib``) and raises ImportError if the
library or one of its entry points is missing.  Only ``mat_mul_amd.build`` (which compiles that
library) and ``mat_mul_amd.sharding`` can be imported without it, which is why attributes are
resolved lazily.
"""
import importlib

__version__ = "0.1.0"
__all__ = ["TensorGameEnv", "SyntheticDemos", "TranspositionTable", "TensorGameError", "functional", "ops", "demo_io",
           "shard_range"]

_SUBMODULES = {"_lib", "ops", "functional", "env", "generator", "sharding", "demo_io", "build", "tree"}
_ATTRS = {
    "TensorGameEnv": "env",
    "SyntheticDemos": "generator",
    "TranspositionTable": "tree",
    "TensorGameError": "_lib",
    "shard_range": "sharding",
}


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module(f"{__name__}.{name}")
    if name in _ATTRS:
        return getattr(importlib.import_module(f"{__name__}.{_ATTRS[name]}"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(set(globals()) | _SUBMODULES | set(_ATTRS))
