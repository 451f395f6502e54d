"""``SyntheticDemos``: the synthetic-demonstration generator on the GPU.

Mirrors ``SyntheticDemoDataset`` (reference datasets.py:20-158) for the arithmetic it does --
``_create_synthetic_demos`` (:124-142), ``_factor_sample`` (:155-158), ``_take_actions``
(:144-153) and the state/action framing of ``__getitem__`` (:84-122) -- with the demos held as
packed int8 arrays in HBM instead of two pickles per demo on disk.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from ._lib import TensorGameError


class SyntheticDemos:
    """``n_demos`` demonstrations of ``max_actions`` rank-1 terms each, dimension ``dim_3d``.

    Same constructor vocabulary as the reference (datasets.py:23-36): max_actions, n_demos,
    dim_t, dim_3d, device, values, probs, shift.  ``seed`` + ``game_id_offset`` key the
    counter-based RNG by GLOBAL demo id, so any sharding of the demos gives identical bytes.
    """

    def __init__(self, max_actions: int, n_demos: int, dim_t: int, dim_3d: int, device="cuda",
                 values=(-1, 0, 1), probs=(0.15, 0.7, 0.15), shift: int = 1, seed: int = 0,
                 game_id_offset: int = 0, basis: Optional[torch.Tensor] = None):
        self.max_actions, self.n_demos, self.dim_t, self.dim_3d = max_actions, n_demos, dim_t, dim_3d
        self.shift, self.seed, self.game_id_offset = shift, seed, game_id_offset
        self.device = torch.device(device)
        self.overflow = torch.zeros((n_demos,), dtype=torch.uint8, device=self.device)
        self.action_seq, self.target_tensor = ops.gen_demos(
            n_demos, dim_3d, max_actions, self.device, values=values, probs=probs, shift=shift,
            seed=seed, game_id_offset=game_id_offset, basis=basis, overflow=self.overflow)

    @classmethod
    def generate(cls, n_demos: int, dim_3d: int, max_actions: int, device="cuda", dim_t: int = 1, **kw) -> "SyntheticDemos":
        """``SyntheticDemos.generate(...)`` of SURVEY.md section 8(b): the same as the constructor, with the
        three sizes first.  ``random_basis=True`` draws a GL(S,Z) basis per demo (``ops.sample_basis``) keyed like
        the demos themselves and emits every demo in it (BASELINE config 5)."""
        if kw.pop("random_basis", False):
            kw["basis"] = ops.sample_basis(n_demos, dim_3d, device, seed=kw.get("seed", 0) ^ 0x5EED,
                                           game_id_offset=kw.get("game_id_offset", 0))
        return cls(max_actions, n_demos, dim_t, dim_3d, device=device, **kw)

    @classmethod
    def sharded(cls, max_actions: int, n_demos_global: int, dim_t: int, dim_3d: int, rank: int, world_size: int,
                device="cuda", **kw):
        """The demos [lo, hi) of an ``n_demos_global`` dataset owned by ``rank`` (contiguous range; the
        generator is keyed by global demo id, so the shards concatenate to the single-rank dataset)."""
        from .sharding import shard_range

        lo, hi = shard_range(n_demos_global, rank, world_size)
        return cls(max_actions, hi - lo, dim_t, dim_3d, device=device, game_id_offset=lo, **kw)

    def __len__(self) -> int:
        return self.n_demos * self.max_actions

    def save(self, path) -> None:
        """One packed int8 file for the whole dataset (demo_io.save_packed)."""
        from . import demo_io

        demo_io.save_packed(path, self.action_seq, self.target_tensor.contiguous(), self.shift, self.seed,
                            self.game_id_offset)

    def export_reference_layout(self, save_dir) -> int:
        """Write ``action_seq_{i}.pt`` / ``target_tensor_{i}.pt`` exactly as the reference does
        (datasets.py:62-69), so its SyntheticDemoDataset(overwrite=False, save_dir=...) reads them."""
        from . import demo_io

        return demo_io.export_reference_layout(save_dir, self.action_seq, self.target_tensor, self.game_id_offset)

    def take_actions(self, idx_action: int) -> torch.Tensor:
        """State of every demo after un-doing the actions that FOLLOW ``idx_action``
        (datasets.py:90-92): target - sum_{j > idx} tensor(a_j)."""
        if not (0 <= idx_action < self.max_actions):
            raise TensorGameError("take_actions", -1, "idx_action out of range")
        if idx_action == self.max_actions - 1:
            return self.target_tensor.clone()
        out, _ = ops.step_many(self.target_tensor, self.action_seq[:, idx_action + 1:].contiguous(), shift=self.shift)
        return out

    def batch(self, idx_action: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """The ``__getitem__`` tuple (datasets.py:84-122) for EVERY demo at one action index:
        (state int8 (n_demos,dim_t,S,S,S), scalar fp32 (n_demos,1), action int8 (n_demos,3S),
        reward fp32 (n_demos,1))."""
        S, T, R = self.dim_3d, self.dim_t, self.max_actions
        head = self.take_actions(idx_action)
        frames = [head]
        for j in reversed(range(idx_action + 1, min(idx_action + T, R))):  # datasets.py:97-102
            frames.append(ops.gen_from_factors(self.action_seq[:, j:j + 1].contiguous(), S, shift=self.shift))
        while len(frames) < T:  # zero padding, datasets.py:105-114
            frames.append(torch.zeros_like(head))
        state = torch.stack([f.reshape(self.n_demos, S, S, S) for f in frames], dim=1)
        scalar = torch.full((self.n_demos, 1), float(R - idx_action), device=self.device)
        reward = torch.full((self.n_demos, 1), float(-(idx_action + 1)), device=self.device)
        return state, scalar, self.action_seq[:, idx_action], reward
