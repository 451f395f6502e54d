"""``TensorGameEnv``: the batched env.reset()/env.step() surface.

The reference has no env class: *reset* is ``build_matmul_tensor``
(utils.py:143-161) or a synthetic start tensor (training.py:363-392), *step* is
``get_child_states`` (act.py:266-275) and *done* is ``tensor_factorized`` (utils.py:181-188).
This class is those three functions over a batch of B independent games resident in HBM as
int8, one HIP kernel launch per call.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import ops
from ._lib import TensorGameError
from .sharding import shard_range


class TensorGameEnv:
    """B independent tensor games of size S x S x S on one MI355X.

    Args mirror the reference's flags: ``dim_3d`` (training.py:83), ``shift`` (utils.py:56).
    ``game_id_offset`` is the global id of local game 0 (sharded runs, section 8e).
    """

    def __init__(self, batch_size: int, dim_3d: int, device="cuda", shift: int = 1,
                 track_overflow: bool = True, game_id_offset: int = 0):
        self.B, self.S, self.shift = int(batch_size), int(dim_3d), int(shift)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TensorGameError("TensorGameEnv", -1, "a ROCm device is required; there is no CPU path")
        self.game_id_offset = int(game_id_offset)
        self.state = ops.alloc_states(self.B, self.S, self.device)
        self.done = torch.zeros((self.B,), dtype=torch.uint8, device=self.device)
        self.overflow = torch.zeros((self.B,), dtype=torch.uint8, device=self.device) if track_overflow else None
        self.t = 0

    @classmethod
    def sharded(cls, global_batch: int, dim_3d: int, rank: int, world_size: int, device, **kw):
        """The shard of a ``global_batch``-game env owned by ``rank`` (contiguous game range)."""
        lo, hi = shard_range(global_batch, rank, world_size)
        return cls(hi - lo, dim_3d, device=device, game_id_offset=lo, **kw)

    # -- reset ------------------------------------------------------------------------------
    def reset(self, start: Optional[torch.Tensor] = None) -> torch.Tensor:
        """start=None: every game <- the <n,n,n> matmul tensor, n = sqrt(dim_3d)
        (datasets.py:273-277).  start (S,S,S): broadcast.  start (B,S,S,S): copied."""
        if start is None:
            n = math.isqrt(self.S)
            if n * n != self.S:
                raise TensorGameError("reset", -1, f"dim_3d={self.S} is not a perfect square; pass a start tensor")
            ops.reset_matmul(self.state, n)
        else:
            start = torch.as_tensor(start).to(device=self.device, dtype=torch.int8)
            if start.dim() == 3:
                ops.reset_broadcast(self.state, start)
            elif tuple(start.shape) == tuple(self.state.shape):
                self.state.copy_(start)
            else:
                raise TensorGameError("reset", -1, f"start must be (S,S,S) or (B,S,S,S), got {tuple(start.shape)}")
        self.done.zero_()
        if self.overflow is not None:
            self.overflow.zero_()
        self.t = 0
        return self.state

    # -- step -------------------------------------------------------------------------------
    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """One action per game, in place.  actions: int8 tokens (B,3S) (other integer dtypes are
        converted with a range check).  Returns (state, done)."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        ops.step(self.state, actions, out=self.state, done=self.done, overflow=self.overflow, shift=self.shift)
        self.t += 1
        return self.state, self.done

    def step_many(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """K actions per game in one launch (state stays on chip).  Returns (state, done_step)."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        _, done_step = ops.step_many(self.state, actions, out=self.state, overflow=self.overflow, shift=self.shift)
        self.t += actions.shape[1]
        return self.state, done_step

    def expand(self, actions: torch.Tensor):
        """k candidate children per game (the env is not advanced).  Returns (children, done, changed)."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        return ops.expand(self.state, actions, shift=self.shift)

    def nnz(self) -> torch.Tensor:
        return ops.done(self.state, want_nnz=True)[1]

    def any_overflow(self) -> bool:
        return bool(self.overflow.any()) if self.overflow is not None else False
