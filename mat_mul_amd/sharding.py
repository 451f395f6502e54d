"""Game-range sharding: games are independent, so N GPUs each own a
contiguous range of GLOBAL game ids.  No collective is involved anywhere on the path."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_games: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[lo, hi) of the global game ids owned by ``rank``: contiguous, sizes differ by at most 1,
    the union over ranks is [0, n_games) with no overlap."""
    if world_size < 1 or not (0 <= rank < world_size) or n_games < 0:
        raise ValueError(f"bad shard request: n_games={n_games} rank={rank} world_size={world_size}")
    base, rem = divmod(n_games, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RankGroup:
    """The control plane of a sharded run: one process per GPU, no collective on the data path.
    The only communication is a barrier and a max-reduction of host-side scalars (elapsed time,
    failure flags) -- over RCCL ("nccl") on GPUs, over "gloo" in CPU tests."""

    def __init__(self, backend: str = "nccl", device=None):
        import os

        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            if not dist.is_initialized():
                kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
                dist.init_process_group(backend, **kw)
            self.dist = dist

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, *values: float):
        """Element-wise MAX of host scalars over all ranks (identity when world == 1)."""
        if self.dist is None:
            return tuple(float(v) for v in values)
        import torch

        on_gpu = self.device is not None and self.dist.get_backend() == "nccl"
        t = torch.tensor(values, dtype=torch.float64, device=self.device if on_gpu else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    def close(self) -> None:
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
            self.dist = None
