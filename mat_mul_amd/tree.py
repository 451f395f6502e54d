"""``TranspositionTable``: the key set of the reference's search tree, resident in HBM.

extend_tree keeps ``new_mc_tree`` -- a dict keyed by ``state_to_str`` strings (utils.py:164-169) -- and uses its keys
twice per expansion: candidates already in the tree are dropped (``c not in new_mc_tree``, act.py:188-195) and the
expanded state's key is recorded (act.py:209-211).  Here the keys are the 64-bit keys of ``ops.state_hash`` /
``ops.expand(want_keys=True)`` and the set is an open-addressing table on the device (``tg_seen_u64``), so a whole batch
of expansions is filtered by one call.  The tree's *values* (children lists, visit counts, q values) are search
bookkeeping and stay with the caller; this class only answers membership.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import TensorGameError


class TranspositionTable:
    def __init__(self, capacity: int, device="cuda"):
        """``capacity`` slots (a power of two); keep the number of recorded keys at or below half of it."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TensorGameError("TranspositionTable", -1, "a ROCm device is required; there is no CPU path")
        self.table = ops.alloc_seen_table(capacity, self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)

    @property
    def capacity(self) -> int:
        return self.table.numel()

    def fresh(self, keys: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
        """uint8, shape of ``keys``: 1 where mask is set and the key is not in the table (act.py:192-194 with
        ``mask`` = the `changed` flags of the expansion, i.e. after remove_null_actions)."""
        return ops.seen(keys.contiguous(), self.table, mask=mask, insert=False)

    def insert(self, keys: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
        """Records the (masked) keys (act.py:209-211); returns which of them were new before this call."""
        return ops.seen(keys.contiguous(), self.table, mask=mask, insert=True, status=self.status)

    def full(self) -> bool:
        """True once a key could not be recorded (synchronises)."""
        return bool(int(self.status[0]) & 1)

    def count(self) -> int:
        """Number of recorded keys (synchronises; a debugging aid, not for hot loops)."""
        return int((self.table != 0).sum())

    def clear(self) -> None:
        self.table.zero_()
        self.status.zero_()
