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
    ``track_nnz``: True / False force, None (default) = by batch size (``TRACKED_FROM``): the env then carries every
    game's count of non-zero entries and ``step()`` loads only the rows an action touches.  The count belongs to the
    env: code that writes ``env.state`` directly (instead of ``reset()``) must call ``recount()`` afterwards.
    """

    TRACKED_FROM = {16: 12000, 25: 2048}  # games from which step() takes tg_step_tracked_i8 when track_nnz is left to the env

    def __init__(self, batch_size: int, dim_3d: int, device="cuda", shift: int = 1,
                 track_overflow: bool = True, game_id_offset: int = 0, dim_t: int = 1, track_nnz: Optional[bool] = None):
        self.B, self.S, self.shift = int(batch_size), int(dim_3d), int(shift)
        self.T = int(dim_t)  # history depth (reference --dim_t, training.py:75); 1 = head only
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TensorGameError("TensorGameEnv", -1, "a ROCm device is required; there is no CPU path")
        self.game_id_offset = int(game_id_offset)
        # T frame slots per game; the head lives in slot self.head and every step writes the next
        # slot, so the reference's history shift (act.py:271-274) is a pointer bump, not a copy
        self.ring = ops.alloc_ring(self.B, self.S, self.T, self.device)
        self.head = 0
        self.done = torch.zeros((self.B,), dtype=torch.uint8, device=self.device)
        self.overflow = torch.zeros((self.B,), dtype=torch.uint8, device=self.device) if track_overflow else None
        self.t = 0
        # track_nnz (dim_t == 1): the env carries every game's number of non-zero entries and steps with
        # ``tg_step_tracked_i8``, which loads only the rows an action touches (S = 16, 25: 1.1-1.9x from a few thousand games
        # on); ``nnz()`` then costs nothing.  Whoever writes ``state`` behind the env's back must call ``recount()``.
        # track_nnz=None (default) decides by the measured crossovers: the tracked step pays two
        # dependent round trips for ~30 % of the lines, so it wins once the batch no longer fits one round of wavefronts --
        # S = 16 from 12 000 games on (8 192: 5.96 against 5.39 us; 12 288: 8.4 against 11.1), S = 25 from 2 048 (10.0
        # against 10.7; 4 096: 12.5 against 13.8; 2 GiB of states: 300 against 574) -- and only when dim_t == 1.
        if track_nnz and self.T != 1:
            raise TensorGameError("TensorGameEnv", -1, "track_nnz needs dim_t == 1 (the tracked step is in place)")
        if track_nnz is None:
            track_nnz = self.T == 1 and self.B >= self.TRACKED_FROM.get(self.S, 1 << 62)
        self._nnz = torch.zeros((self.B,), dtype=torch.int32, device=self.device) if track_nnz else None

    @property
    def state(self) -> torch.Tensor:
        """The current head state, int8 (B,S,S,S) (a view of the ring's head slot)."""
        return self.ring[:, self.head]

    @classmethod
    def sharded(cls, global_batch: int, dim_3d: int, rank: int, world_size: int, device, **kw):
        """The shard of a ``global_batch``-game env owned by ``rank`` (contiguous game range)."""
        lo, hi = shard_range(global_batch, rank, world_size)
        return cls(hi - lo, dim_3d, device=device, game_id_offset=lo, **kw)

    # -- reset ------------------------------------------------------------------------------
    def reset(self, start: Optional[torch.Tensor] = None) -> torch.Tensor:
        """start=None: every game <- the <n,n,n> matmul tensor, n = sqrt(dim_3d)
        (datasets.py:273-277).  start (S,S,S): broadcast.  start (B,S,S,S): copied."""
        if self.T > 1:
            self.ring.zero_()  # history frames of a fresh game are zero (build_matmul_tensor, utils.py:157)
        self.head = 0
        if start is not None and torch.as_tensor(start).dim() == 5:  # a full (B,T,S,S,S) history, newest first
            start = torch.as_tensor(start).to(device=self.device, dtype=torch.int8)
            if tuple(start.shape) != tuple(self.ring.shape):
                raise TensorGameError("reset", -1, f"history must be {tuple(self.ring.shape)}")
            for f in range(self.T):  # frame f (f steps old) -> slot (head - f) mod T
                self.ring[:, (self.head - f) % self.T].copy_(start[:, f])
        elif start is None:
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
        if self._nnz is not None:
            self.recount()
        return self.state

    def recount(self) -> None:
        """(track_nnz) count the non-zero entries of every game again -- after a reset, or after the state was written from
        outside the env."""
        if self._nnz is not None:
            self._nnz.copy_(ops.done(self.state, want_nnz=True)[1])

    # -- step -------------------------------------------------------------------------------
    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """One action per game, in place.  actions: int8 tokens (B,3S) (other integer dtypes are
        converted with a range check).  Returns (state, done)."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        if self._nnz is not None:  # in place, only the touched rows loaded, the count carried
            ops.step_tracked(self.state, actions, self._nnz, done=self.done, overflow=self.overflow, shift=self.shift)
            self.t += 1
            return self.state, self.done
        nxt = (self.head + 1) % self.T
        ops.step(self.ring[:, self.head], actions, out=self.ring[:, nxt], done=self.done, overflow=self.overflow,
                 shift=self.shift)
        self.head = nxt
        self.t += 1
        return self.state, self.done

    def step_observe(self, actions: torch.Tensor, dtype=torch.float32, out=None, scalars=None):
        """``step()`` and ``model_input()`` of the new state in one call (one kernel at S=4): what the tree search does
        between two network evaluations (act.py:178-183).  Returns (model_state (B,T,S,S,S), scalars (B,1), done)."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        x, sc, _, nxt = ops.step_emit(self.ring, self.head, actions, float(self.t + 1), dtype=dtype, out=out, scalars=scalars,
                                      done=self.done, overflow=self.overflow, shift=self.shift)
        self.head = nxt
        self.t += 1
        if self._nnz is not None:
            self.recount()
        return x, sc, self.done

    def graph_stepper(self, actions: torch.Tensor):
        """``step()`` bound to a STATIC token buffer and replayed as a hipGraph: returns a callable that applies
        whatever tokens ``actions`` (int8 (B,3S) on this device; refill it in place between calls) holds at that
        moment.  A replay costs the launch boundary (about 2.5 us at S=4, B=65 536) instead of the ~10 us of Python
        and ctypes in ``step()``; one graph per history slot is captured on first use."""
        if actions.dtype != torch.int8 or tuple(actions.shape) != (self.B, 3 * self.S) or actions.device != self.device \
                or not actions.is_contiguous():
            raise TensorGameError("graph_stepper", -1, f"actions must be contiguous int8 {(self.B, 3 * self.S)} on {self.device}")
        graphs = {}
        side = torch.cuda.Stream(device=self.device)
        calls = [0]

        def step() -> Tuple[torch.Tensor, torch.Tensor]:
            slot = self.head
            nxt = (slot + 1) % self.T
            # Two graphs per slot, captured back to back and replayed in turn: consecutive tg_step_i8 launches sweep the batch
            # in opposite directions (the tail of one sweep is the head of the next in L2 / the Infinity Cache), and a
            # captured launch keeps the direction it was captured with.
            parity = calls[0] & 1
            calls[0] += 1
            g = graphs.get((slot, parity))
            if g is None:
                cur = torch.cuda.current_stream(self.device)
                side.wait_stream(cur)
                for p in (0, 1):
                    gp = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gp, stream=side):  # capture only records the launch: nothing runs here
                        if self._nnz is not None:
                            ops.step_tracked(self.state, actions, self._nnz, done=self.done, overflow=self.overflow, shift=self.shift)
                        else:
                            ops.step(self.ring[:, slot], actions, out=self.ring[:, nxt], done=self.done,
                                     overflow=self.overflow, shift=self.shift)
                    graphs[(slot, p)] = gp
                cur.wait_stream(side)
                g = graphs[(slot, parity)]
            g.replay()
            self.head = nxt
            self.t += 1
            return self.state, self.done

        return step

    def step_many(self, actions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """K actions per game in one launch (state stays on chip).  Returns (state, done_step)."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        if self.T > 1:
            raise TensorGameError("step_many", -1, "step_many keeps no history; use step() when dim_t > 1")
        _, done_step = ops.step_many(self.state, actions, out=self.state, overflow=self.overflow, shift=self.shift)
        self.t += actions.shape[1]
        if self._nnz is not None:
            self.recount()
        return self.state, done_step

    def step_stream(self, actions: torch.Tensor, ready=None, progress=None, status=None,
                    done: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """K steps in ONE resident launch (``tg_step_stream_i8``; S = 4, 16, 25, dim_t == 1): actions int8 (K,B,3S) step-major,
        consumed as ``ready[k]`` (device int32 (K), or None = all valid now) is set by a producer on the same device -- the
        policy network choosing action k from state k (act.py:182-183).  Same states and ``done[k]`` as K calls of
        ``step()``, without the launch boundary between them (S=4, B=65 536: 0.5 us per step against 2.3).
        Returns (state, done (K,B)); ``self.done`` is the last step's."""
        if self.T > 1:
            raise TensorGameError("step_stream", -1, "the streamed stepper keeps no history; use step() when dim_t > 1")
        if actions.dtype != torch.int8:
            raise TensorGameError("step_stream", -1, "actions must be int8 tokens (K,B,3S), step-major (ops.as_tokens)")
        _, dn = ops.step_stream(self.state, actions, done=done, overflow=self.overflow, ready=ready, progress=progress,
                                status=status, shift=self.shift)
        self.done.copy_(dn[-1])
        self.t += actions.shape[0]
        if self._nnz is not None:
            self.recount()
        return self.state, dn

    def expand(self, actions: torch.Tensor, want_keys: bool = False):
        """k candidate children per game (the env is not advanced).  Returns (children, done, changed[, keys])."""
        if actions.dtype != torch.int8:
            actions = ops.as_tokens(actions, self.device)
        return ops.expand(self.state, actions, shift=self.shift, want_keys=want_keys)

    def snapshot(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A copy of the current head states (the reference's step is functional: callers such as the tree search
        keep the parent, act.py:183-195; here the step is in place, so keeping a parent is an explicit copy)."""
        return ops.copy_states(self.state, out)

    def model_input(self, dtype=torch.float32):
        """(state (B,T,S,S,S) float, scalars (B,1)) as AlphaTensor.fwd_* consume them
        (model.py:101-122): newest frame first, scalars = the time step (utils.py:22-37)."""
        return ops.emit_frames(self.ring, self.head, float(self.t), dtype=dtype)

    def hash(self) -> torch.Tensor:
        return ops.state_hash(self.state)

    def rank_reward(self) -> torch.Tensor:
        """-sum of slice ranks per game: the terminal reward of act.py:59 / :214."""
        return -ops.slice_rank(self.state)

    def nnz(self) -> torch.Tensor:
        if self._nnz is not None:
            return self._nnz
        return ops.done(self.state, want_nnz=True)[1]

    def any_overflow(self) -> bool:
        return bool(self.overflow.any()) if self.overflow is not None else False
