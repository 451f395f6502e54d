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
