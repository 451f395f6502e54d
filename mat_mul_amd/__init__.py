"""mat_mul_amd -- MI355X-native batched tensor-decomposition environment.

The hot path of kurtosis/mat_mul (AlphaTensor re-implementation) -- the per-game state update
``state <- state - u(x)v(x)w`` with the all-zero terminal check, and the synthetic-demonstration
generator -- as hand-written HIP kernels for gfx950 behind the C ABI of ``include/tensor_game.h``.

Importing this package loads ``mat_mul_amd/lib/libtensorgame.so`` and fails loudly if it is
missing: there is no CPU fallback.
"""
from . import _lib, functional, ops
from ._lib import TensorGameError
from .env import TensorGameEnv
from .generator import SyntheticDemos
from .sharding import shard_range

__all__ = ["TensorGameEnv", "SyntheticDemos", "TensorGameError", "functional", "ops", "shard_range"]
__version__ = "0.1.0"
