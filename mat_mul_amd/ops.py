"""Tensor-level wrappers over the C ABI (include/tensor_game.h).

Every function takes PyTorch-ROCm tensors, checks device / dtype / layout on the host,
and enqueues ONE kernel on the caller's current HIP stream (so calls can be captured in a
``torch.cuda.graph``).  PyTorch is plumbing here: device memory and streams.  There is
no CPU path: a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import TG_MAX_ACTIONS, TG_MAX_S, TG_MAX_VALUES, TensorGameError, call

__all__ = [
    "step", "step_tracked", "copy_states", "prepare_step", "step_many", "step_stream", "step_stream_layout", "step_stream_capacity", "expand", "done", "reset_matmul", "reset_broadcast", "gen_from_factors",
    "gen_demos", "sample_basis", "change_basis", "as_tokens", "categorical_thresholds",
    "alloc_states", "alloc_ring", "emit_frames", "step_emit", "state_hash", "slice_rank", "alloc_seen_table", "seen",
]


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream(dev: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise TensorGameError(name, -1, f"{name} must live on a ROCm device (got {t.device}); there is no CPU path")


def _state_layout(state: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """(B, S, game_stride_bytes) of an int8 (B,S,S,S) tensor whose games are dense."""
    _need_gpu(state, name)
    if state.dtype != torch.int8 or state.dim() != 4 or not (state.shape[1] == state.shape[2] == state.shape[3]):
        raise TensorGameError(name, -1, f"{name} must be int8 of shape (B,S,S,S), got {state.dtype} {tuple(state.shape)}")
    B, S = state.shape[0], state.shape[1]
    if B > 0 and state.stride()[1:] != (S * S, S, 1):
        raise TensorGameError(name, -1, f"{name}: each game must be C-contiguous (S,S,S)")
    stride = state.stride(0) if B > 1 else max(state.stride(0), S ** 3)
    if stride < S ** 3:
        raise TensorGameError(name, -1, f"{name}: game stride {stride} < S^3")
    return B, S, stride


def _tokens(actions: torch.Tensor, lead: Tuple[int, ...], S: int, dev: torch.device, name: str) -> torch.Tensor:
    _need_gpu(actions, name)
    if actions.dtype != torch.int8:
        raise TensorGameError(name, -1, f"{name} must be int8 tokens (use ops.as_tokens), got {actions.dtype}")
    if tuple(actions.shape) != (*lead, 3 * S):
        raise TensorGameError(name, -1, f"{name} must have shape {(*lead, 3 * S)}, got {tuple(actions.shape)}")
    if actions.device != dev:
        raise TensorGameError(name, -1, f"{name} is on {actions.device}, state on {dev}")
    return actions if actions.is_contiguous() else actions.contiguous()


def _flag(t: Optional[torch.Tensor], shape: Tuple[int, ...], dtype, dev, name: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != dtype or tuple(t.shape) != shape or t.device != dev or not t.is_contiguous():
        raise TensorGameError(name, -1, f"{name} must be contiguous {dtype} {shape} on {dev}")
    return t


def as_tokens(actions, device=None, check: bool = True) -> torch.Tensor:
    """int64 (reference dtype, datasets.py:140) or any integer tensor -> int8 tokens.
    ``check`` verifies the values fit int8 (one host sync); the reference never range-checks
    tokens (utils.py:64-66), the build refuses values it cannot represent."""
    t = torch.as_tensor(actions)
    if t.dtype != torch.int8:
        if t.is_floating_point():
            raise TensorGameError("as_tokens", -1, "tokens must be integers")
        if check and t.numel() and (int(t.min()) < -128 or int(t.max()) > 127):
            raise TensorGameError("as_tokens", -1, "token outside int8 range")
        t = t.to(torch.int8)
    if device is not None:
        t = t.to(device)
    return t.contiguous()


def alloc_states(B: int, S: int, device, pad_to: int = 16, zero: bool = True) -> torch.Tensor:
    """int8 (B,S,S,S) states whose game stride is S^3 rounded up to ``pad_to`` bytes (16 keeps every
    game on the dwordx4 fast path; S=25 -> 15632).  Zero-filled unless ``zero=False`` (outputs that a
    kernel overwrites completely; the padding bytes are never read)."""
    n = S ** 3
    stride = -(-n // pad_to) * pad_to
    buf = (torch.zeros if zero else torch.empty)((B, stride), dtype=torch.int8, device=device)
    return buf[:, :n].unflatten(1, (S, S, S))


def step(state, actions, out=None, done=None, overflow=None, shift: int = 1):
    """state_out = state - u(x)v(x)w(actions); done[b] = state_out[b] all zero.
    == reference get_child_states (act.py:266-275, k=1,T=1) + tensor_factorized per game
    (utils.py:181-188).  ``out=state`` steps in place.  Returns (state_out, done)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    actions = _tokens(actions, (B,), S, dev, "actions")
    if out is None:
        out = torch.empty_strided(state.shape, state.stride(), dtype=torch.int8, device=dev)
    Bo, So, ostride = _state_layout(out, "out")
    if (Bo, So) != (B, S) or (B > 1 and ostride != stride) or out.device != dev:
        raise TensorGameError("step", -1, "out must match state's shape, stride and device")
    if done is None:
        done = torch.empty((B,), dtype=torch.uint8, device=dev)
    done = _flag(done, (B,), torch.uint8, dev, "done")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    with torch.cuda.device(dev):
        call("tg_step_i8", _ptr(state), _ptr(out), _ptr(actions), _ptr(done), _ptr(overflow),
             B, S, stride, int(shift), _stream(dev))
    return out, done


def copy_states(state, out=None):
    """out[b] = state[b] (a snapshot of a batch of games; the reference's step is functional and its callers
    keep the parent, act.py:183-195).  ``out`` may have a different game stride.  Returns out."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    if out is None:
        out = alloc_states(B, S, dev, zero=False)
    Bo, So, ostride = _state_layout(out, "out")
    if (Bo, So) != (B, S) or out.device != dev:
        raise TensorGameError("copy_states", -1, "out must match state's shape and device")
    with torch.cuda.device(dev):
        call("tg_copy_i8", _ptr(state), _ptr(out), B, S, stride, ostride, _stream(dev))
    return out


def prepare_step(state, actions_seq, done, overflow=None, shift: int = 1):
    """Validate once, launch many: returns ``launch(k)`` that enqueues one in-place
    ``tg_step_i8`` with ``actions_seq[k]`` on the current stream with no per-call checks
    (for rollouts and hipGraph capture, where Python overhead would dominate a 2 us kernel)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    acts = [_tokens(a, (B,), S, dev, "actions") for a in actions_seq]
    done = _flag(done, (B,), torch.uint8, dev, "done")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    fn = _lib.lib.tg_step_i8
    sp, dp, op = _ptr(state), _ptr(done), _ptr(overflow)
    aps = [_ptr(a) for a in acts]
    Bc, Sc, stc, shc = C.c_int64(B), C.c_int(S), C.c_int64(stride), C.c_int(int(shift))
    index = dev.index if dev.index is not None else torch.cuda.current_device()

    def launch(k: int) -> None:
        rc = fn(sp, sp, aps[k], dp, op, Bc, Sc, stc, shc, C.c_void_p(torch.cuda.current_stream(index).cuda_stream))
        if rc != 0:
            raise TensorGameError("tg_step_i8", rc, _lib.lib.tg_last_error().decode())

    launch.keepalive = (state, acts, done, overflow)  # the raw pointers above must stay valid
    return launch


def step_many(state, actions, out=None, done_step=None, overflow=None, shift: int = 1):
    """K sequential steps, state resident on chip.  actions int8 (B,K,3S).
    == reference SyntheticDemoDataset._take_actions (datasets.py:144-153).
    Returns (state_out, done_step int32: first step whose post-state is zero, or -1)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    if actions.dim() != 3:
        raise TensorGameError("step_many", -1, "actions must be (B,K,3S)")
    K = actions.shape[1]
    actions = _tokens(actions, (B, K), S, dev, "actions")
    if out is None:
        out = torch.empty_strided(state.shape, state.stride(), dtype=torch.int8, device=dev)
    Bo, So, ostride = _state_layout(out, "out")
    if (Bo, So) != (B, S) or (B > 1 and ostride != stride) or out.device != dev:
        raise TensorGameError("step_many", -1, "out must match state's shape, stride and device")
    if done_step is None:
        done_step = torch.empty((B,), dtype=torch.int32, device=dev)
    done_step = _flag(done_step, (B,), torch.int32, dev, "done_step")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    with torch.cuda.device(dev):
        call("tg_step_many_i8", _ptr(state), _ptr(out), _ptr(actions), _ptr(done_step), _ptr(overflow),
             B, S, K, stride, int(shift), _stream(dev))
    return out, done_step


def step_tracked(state, actions, nnz, done=None, overflow=None, shift: int = 1):
    """The in-place step that reads only what the action touches: ``nnz`` int32 (B,) carries the exact number of non-zero
    entries per game (``done(state)[1]`` computes it) and is updated; ``done[b] = (nnz[b] == 0)``.  Same state, done and
    overflow as ``step(state, actions, out=state)``.  Returns (state, done)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    actions = _tokens(actions, (B,), S, dev, "actions")
    if nnz.dtype != torch.int32 or tuple(nnz.shape) != (B,) or not nnz.is_contiguous() or nnz.device != dev:
        raise TensorGameError("step_tracked", -1, f"nnz must be a contiguous int32 ({B},) tensor on {dev}")
    if done is None:
        done = torch.empty((B,), dtype=torch.uint8, device=dev)
    done = _flag(done, (B,), torch.uint8, dev, "done")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    with torch.cuda.device(dev):
        call("tg_step_tracked_i8", _ptr(state), _ptr(actions), _ptr(nnz), _ptr(done), _ptr(overflow), B, S, stride, int(shift),
             _stream(dev))
    return state, done


def step_stream_layout(B: int, S: int, device=None) -> Tuple[int, int]:
    """(n_units, games_per_unit) of ``step_stream``: unit u (a wavefront) owns games [u*gpu, (u+1)*gpu)."""
    n, g = C.c_int64(0), C.c_int(0)
    with torch.cuda.device(torch.device(device) if device is not None else torch.cuda.current_device()):
        call("tg_step_stream_layout", B, S, C.byref(n), C.byref(g))
    return int(n.value), int(g.value)


def step_stream_capacity(S: int, device=None) -> int:
    """The largest batch ``step_stream`` takes together with ready words on this device (every unit resident at once)."""
    n = C.c_int64(0)
    with torch.cuda.device(torch.device(device) if device is not None else torch.cuda.current_device()):
        call("tg_step_stream_capacity", S, C.byref(n))
    return int(n.value)


def step_stream(state, actions, done=None, overflow=None, ready=None, progress=None, status=None, shift: int = 1):
    """K in-place steps in ONE launch for action blocks that become available step by step.
    actions int8 (K,B,3S) STEP-major; ready uint32/int32 (K) (step k waits for ready[k] != 0) or None;
    done uint8 (K,B); progress int32 (n_units); status int32 (1).  == K calls of ``step(state, actions[k], out=state)``
    without the launch boundary between them.  Returns (state, done)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    if actions.dim() != 3:
        raise TensorGameError("step_stream", -1, "actions must be (K,B,3S), step-major")
    K = actions.shape[0]
    actions = _tokens(actions, (K, B), S, dev, "actions")
    if done is None:
        done = torch.empty((K, B), dtype=torch.uint8, device=dev)
    done = _flag(done, (K, B), torch.uint8, dev, "done")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    for name, t, n in (("ready", ready, K), ("progress", progress, None), ("status", status, 1)):
        if t is not None and (t.dtype not in (torch.int32, torch.uint32) or t.dim() != 1 or not t.is_contiguous()
                              or (n is not None and t.numel() != n) or t.device != dev):
            # (device memory only: the stepper polls and publishes with agent-scope accesses; a host producer
            # releases a step by a fill / copy enqueued on another stream, not by writing mapped memory)
            raise TensorGameError("step_stream", -1, f"{name} must be a contiguous 32-bit vector on {dev}")
    if progress is not None:
        try:
            n_units = step_stream_layout(B, S, dev)[0]
        except TensorGameError:
            if ready is not None or S != 4:
                raise
            n_units = -(-B // 64)  # beyond the resident batch, without ready words: units of 64 games in rounds
        if progress.numel() < n_units:
            raise TensorGameError("step_stream", -1, "progress needs one word per unit (ops.step_stream_layout)")
    with torch.cuda.device(dev):
        call("tg_step_stream_i8", _ptr(state), _ptr(actions), _ptr(done), _ptr(overflow), _ptr(ready), _ptr(progress),
             _ptr(status), B, S, K, stride, int(shift), _stream(dev))
    return state, done


def expand(state, actions, out=None, done=None, changed=None, overflow=None, shift: int = 1, want_keys: bool = False,
           keys=None):
    """k children per parent.  actions int8 (B,k,3S) -> children int8 (B,k,S,S,S), done (B,k),
    changed (B,k).  == reference get_child_states (act.py:266-275) with k>1, T=1, plus the
    per-game form of remove_null_actions (utils.py:191-194).  ``want_keys`` (or a ``keys`` int64 (B,k) tensor) also
    returns the 64-bit key of every child (== ``state_hash`` of the child; the state_to_str keys of act.py:188-190):
    (children, done, changed, keys)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    if actions.dim() != 3:
        raise TensorGameError("expand", -1, "actions must be (B,k,3S)")
    k = actions.shape[1]
    actions = _tokens(actions, (B, k), S, dev, "actions")
    if out is None:
        out = alloc_states(B * k, S, dev, zero=False).unflatten(0, (B, k))
    if out.dtype != torch.int8 or tuple(out.shape) != (B, k, S, S, S) or out.device != dev:
        raise TensorGameError("expand", -1, f"out must be int8 {(B, k, S, S, S)} on {dev}")
    # child (b, i) lives at base + (b*k + i) * ostride: the (B, k) dims must collapse to ONE stride.  No
    # flatten() here -- on a view that cannot be flattened it would silently copy and the kernel would write
    # through the original pointer with the copy's strides
    if out.stride()[2:] != (S * S, S, 1):
        raise TensorGameError("expand", -1, "out: each child must be C-contiguous (S,S,S)")
    ostride = out.stride(1) if k > 1 else (out.stride(0) if B > 1 else max(out.stride(1), S ** 3))
    if ostride < S ** 3 or (B > 1 and k > 1 and out.stride(0) != k * ostride):
        raise TensorGameError("expand", -1, f"out: children must be evenly spaced (strides {out.stride()[:2]}); "
                                            "a slice like big[:, :k] of a wider buffer is not")
    if done is None:
        done = torch.empty((B, k), dtype=torch.uint8, device=dev)
    if changed is None:
        changed = torch.empty((B, k), dtype=torch.uint8, device=dev)
    done = _flag(done, (B, k), torch.uint8, dev, "done")
    changed = _flag(changed, (B, k), torch.uint8, dev, "changed")
    overflow = _flag(overflow, (B, k), torch.uint8, dev, "overflow")
    if want_keys and keys is None:
        keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    if keys is not None:
        keys = _flag(keys, (B, k), torch.int64, dev, "keys")
        with torch.cuda.device(dev):
            call("tg_expand_keyed_i8", _ptr(state), _ptr(out), _ptr(actions), _ptr(done), _ptr(changed),
                 _ptr(overflow), _ptr(keys), B, S, k, stride, ostride, int(shift), _stream(dev))
        return out, done, changed, keys
    with torch.cuda.device(dev):
        call("tg_expand_i8", _ptr(state), _ptr(out), _ptr(actions), _ptr(done), _ptr(changed),
             _ptr(overflow), B, S, k, stride, ostride, int(shift), _stream(dev))
    return out, done, changed


def done(state, want_nnz: bool = False):
    """done[b] = head of game b is all zero (== tensor_factorized, utils.py:181-188, per game);
    nnz[b] = number of non-zero entries (the rank bound of training.py:266)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    d = torch.empty((B,), dtype=torch.uint8, device=dev)
    nnz = torch.empty((B,), dtype=torch.int32, device=dev) if want_nnz else None
    with torch.cuda.device(dev):
        call("tg_done_i8", _ptr(state), _ptr(d), _ptr(nnz), B, S, stride, _stream(dev))
    return (d, nnz) if want_nnz else d


def reset_matmul(out, n: int):
    """every game <- <n,n,n> (== build_matmul_tensor(1,n,n,n)[0], utils.py:143-161)."""
    B, S, stride = _state_layout(out, "out")
    if S != n * n:
        raise TensorGameError("reset_matmul", -1, f"S={S} != n*n={n * n}")
    with torch.cuda.device(out.device):
        call("tg_reset_matmul_i8", _ptr(out), B, int(n), stride, _stream(out.device))
    return out


def reset_broadcast(out, start):
    """every game <- start (int8 (S,S,S) on the same device)."""
    B, S, stride = _state_layout(out, "out")
    _need_gpu(start, "start")
    if start.dtype != torch.int8 or tuple(start.shape) != (S, S, S) or start.device != out.device:
        raise TensorGameError("reset_broadcast", -1, f"start must be int8 {(S, S, S)} on {out.device}")
    start = start.contiguous()
    with torch.cuda.device(out.device):
        call("tg_reset_broadcast_i8", _ptr(start), _ptr(out), B, S, stride, _stream(out.device))
    return out


def gen_from_factors(actions, S: int, out=None, overflow=None, shift: int = 1):
    """target[b] = sum_r tensor(actions[b][r]); actions int8 (B,R,3S).  The deterministic half
    of create_synthetic_demo (utils.py:218-232) / uvw_to_demo (utils.py:40-53)."""
    _need_gpu(actions, "actions")
    if actions.dim() != 3:
        raise TensorGameError("gen_from_factors", -1, "actions must be (B,R,3S)")
    B, R = actions.shape[:2]
    dev = actions.device
    actions = _tokens(actions, (B, R), S, dev, "actions")
    if out is None:
        out = alloc_states(B, S, dev, zero=False)
    Bo, So, stride = _state_layout(out, "out")
    if (Bo, So) != (B, S) or out.device != dev:
        raise TensorGameError("gen_from_factors", -1, "out shape/device mismatch")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    with torch.cuda.device(dev):
        call("tg_gen_from_factors_i8", _ptr(actions), _ptr(out), _ptr(overflow), B, S, R, stride,
             int(shift), _stream(dev))
    return out


def categorical_thresholds(probs: Sequence[float]) -> np.ndarray:
    """uint32 cdf thresholds of a categorical distribution (normalised like
    torch.distributions.Categorical, reference utils.py:198): a 32-bit draw d selects
    values[#{t : d >= t}]."""
    p = np.asarray(probs, dtype=np.float64)
    if p.ndim != 1 or len(p) < 1 or len(p) > TG_MAX_VALUES or (p < 0).any() or p.sum() <= 0:
        raise TensorGameError("categorical_thresholds", -1, f"probs must be 1..{TG_MAX_VALUES} non-negative weights")
    cdf = np.cumsum(p) / p.sum()
    return np.minimum(np.floor(cdf[:-1] * 4294967296.0 + 0.5), 4294967295.0).astype(np.uint32)


def _dist(values, probs, name):
    vals = np.asarray(values, dtype=np.int64)
    if vals.ndim != 1 or len(vals) != len(probs) or (np.abs(vals) > 127).any():
        raise TensorGameError(name, -1, "values must be int8 and match probs")
    thr = categorical_thresholds(probs)
    p = np.asarray(probs, dtype=np.float64)
    if not ((vals != 0) & (p > 0)).any():
        raise TensorGameError(name, -1, "the distribution can never draw a non-zero factor value")
    vals8 = vals.astype(np.int8)
    return (thr, vals8, thr.ctypes.data_as(C.c_void_p), vals8.ctypes.data_as(C.c_void_p), len(vals8))


def gen_demos(B: int, S: int, R: int, device, values=(-1, 0, 1), probs=(0.15, 0.7, 0.15), shift: int = 1,
              seed: int = 0, game_id_offset: int = 0, basis=None, target=None, actions=None, overflow=None):
    """The synthetic-demonstration generator (== create_synthetic_demo, utils.py:203-233 /
    SyntheticDemoDataset._create_synthetic_demos, datasets.py:124-142).  Returns
    (actions int8 (B,R,3S), target int8 (B,S,S,S)).  Deterministic in (seed, global game id)."""
    dev = torch.device(device)
    thr, vals, thr_p, val_p, nv = _dist(values, probs, "gen_demos")
    if target is None:
        target = alloc_states(B, S, dev, zero=False)
    Bo, So, stride = _state_layout(target, "target")
    if actions is None:
        actions = torch.empty((B, R, 3 * S), dtype=torch.int8, device=dev)
    if (Bo, So) != (B, S) or tuple(actions.shape) != (B, R, 3 * S) or actions.dtype != torch.int8 \
            or not actions.is_contiguous() or actions.device != target.device:
        raise TensorGameError("gen_demos", -1, "target/actions shape, dtype or device mismatch")
    dev = target.device
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    if basis is not None:
        if basis.dtype != torch.int8 or tuple(basis.shape) != (B, 3, S, S) or basis.device != dev:
            raise TensorGameError("gen_demos", -1, f"basis must be int8 {(B, 3, S, S)} on {dev}")
        basis = basis.contiguous()
    with torch.cuda.device(dev):
        call("tg_gen_demos_i8", _ptr(target), _ptr(actions), _ptr(overflow), B, S, R, thr_p, val_p, nv,
             int(shift), C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(game_id_offset), _ptr(basis), stride,
             _stream(dev))
    return actions, target


def sample_basis(B: int, S: int, device, values=(-1, 0, 1), probs=None, seed: int = 0,
                 game_id_offset: int = 0, want_factors: bool = False):
    """Three random unimodular matrices per game, P = L @ U (not in the reference).
    ``probs`` are the off-diagonal value weights of L and U; the default keeps about 0.4 non-zero
    off-diagonal entries per row (p = min(0.15, 0.4/S) each for -1 and +1), dense enough to mix
    the basis and sparse enough that transformed int8 targets rarely overflow."""
    dev = torch.device(device)
    if probs is None:
        p = min(0.15, 0.4 / S)
        probs = (p, 1.0 - 2.0 * p, p)
    thr, vals, thr_p, val_p, nv = _dist(values, probs, "sample_basis")
    if (S * vals.astype(np.int64) ** 2 > 127).any():  # the library enforces the same rule
        raise TensorGameError("sample_basis", -1, f"every value needs S*v^2 <= 127 at S={S}: P = L @ U is emitted as int8")
    P = torch.empty((B, 3, S, S), dtype=torch.int8, device=dev)
    L = torch.empty_like(P) if want_factors else None
    U = torch.empty_like(P) if want_factors else None
    with torch.cuda.device(P.device):
        call("tg_sample_basis_i8", _ptr(P), _ptr(L), _ptr(U), B, S, thr_p, val_p, nv,
             C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(game_id_offset), _stream(P.device))
    return (P, L, U) if want_factors else P


def change_basis(state, basis, out=None, overflow=None):
    """T'[a,b,c] = sum A[a,i] B[b,j] C[c,k] T[i,j,k]; basis int32 (B,3,S,S)."""
    B, S, stride = _state_layout(state, "state")
    dev = state.device
    if basis.dtype != torch.int32 or tuple(basis.shape) != (B, 3, S, S) or basis.device != dev:
        raise TensorGameError("change_basis", -1, f"basis must be int32 {(B, 3, S, S)} on {dev}")
    basis = basis.contiguous()
    if out is None:
        out = torch.empty_strided(state.shape, state.stride(), dtype=torch.int8, device=dev)
    Bo, So, ostride = _state_layout(out, "out")
    if (Bo, So) != (B, S) or (B > 1 and ostride != stride):
        raise TensorGameError("change_basis", -1, "out must match state's shape and stride")
    if out.data_ptr() == state.data_ptr():
        raise TensorGameError("change_basis", -1, "in-place change of basis is not supported")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    with torch.cuda.device(dev):
        call("tg_change_basis_i8", _ptr(state), _ptr(basis), _ptr(out), _ptr(overflow), B, S, stride, _stream(dev))
    return out


def alloc_ring(B: int, S: int, T: int, device, pad_to: int = 16) -> torch.Tensor:
    """Zeroed history ring int8 (B,T,S,S,S): T frame slots per game, each padded to ``pad_to`` bytes.
    ``ring[:, s]`` is a valid (B,S,S,S) state for every entry point (game stride = T * frame stride)."""
    n = S ** 3
    fs = -(-n // pad_to) * pad_to
    buf = torch.zeros((B, T, fs), dtype=torch.int8, device=device)
    return buf[:, :, :n].unflatten(2, (S, S, S))


def emit_frames(ring, head_slot: int, t_step: float = 0.0, dtype=torch.float32, out=None, scalars=None):
    """(B,T,S,S,S) model input from the history ring, newest frame first, plus the (B,1) scalars.
    == the history shift of get_child_states (act.py:271-274) + get_scalars (utils.py:22-37)."""
    _need_gpu(ring, "ring")
    if ring.dtype != torch.int8 or ring.dim() != 5 or not (ring.shape[2] == ring.shape[3] == ring.shape[4]):
        raise TensorGameError("emit_frames", -1, f"ring must be int8 (B,T,S,S,S), got {ring.dtype} {tuple(ring.shape)}")
    B, T, S = ring.shape[0], ring.shape[1], ring.shape[2]
    if ring.stride()[2:] != (S * S, S, 1):
        raise TensorGameError("emit_frames", -1, "each frame must be C-contiguous (S,S,S)")
    fs = ring.stride(1) if T > 1 else S ** 3
    gs = ring.stride(0) if B > 1 else max(ring.stride(0), (T - 1) * fs + S ** 3)
    codes = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    if dtype not in codes:
        raise TensorGameError("emit_frames", -1, "dtype must be float32, float16 or bfloat16")
    dev = ring.device
    if out is None:
        out = torch.empty((B, T, S, S, S), dtype=dtype, device=dev)
    if out.dtype != dtype or tuple(out.shape) != (B, T, S, S, S) or not out.is_contiguous() or out.device != dev:
        raise TensorGameError("emit_frames", -1, "out must be contiguous (B,T,S,S,S) of the requested dtype")
    if scalars is None:
        scalars = torch.empty((B, 1), dtype=torch.float32, device=dev)
    scalars = _flag(scalars, (B, 1), torch.float32, dev, "scalars")
    with torch.cuda.device(dev):
        call("tg_emit_frames", _ptr(ring), _ptr(out), _ptr(scalars), codes[dtype], B, S, T,
             int(head_slot) % T, C.c_float(float(t_step)), fs, gs, _stream(dev))
    return out, scalars


def step_emit(ring, head_slot: int, actions, t_step: float = 0.0, dtype=torch.float32, out=None, scalars=None, done=None,
              overflow=None, shift: int = 1):
    """One env step on the history ring and the model input of the new state, in one call (the fused form of
    ``step`` + ``emit_frames``; one kernel at S=4).  The new head is written into slot ``(head_slot + 1) % T``.
    Returns (out (B,T,S,S,S), scalars (B,1), done (B,), new_head_slot)."""
    _need_gpu(ring, "ring")
    if ring.dtype != torch.int8 or ring.dim() != 5 or not (ring.shape[2] == ring.shape[3] == ring.shape[4]):
        raise TensorGameError("step_emit", -1, f"ring must be int8 (B,T,S,S,S), got {ring.dtype} {tuple(ring.shape)}")
    B, T, S = ring.shape[0], ring.shape[1], ring.shape[2]
    if ring.stride()[2:] != (S * S, S, 1):
        raise TensorGameError("step_emit", -1, "each frame must be C-contiguous (S,S,S)")
    fs = ring.stride(1) if T > 1 else S ** 3
    gs = ring.stride(0) if B > 1 else max(ring.stride(0), (T - 1) * fs + S ** 3)
    codes = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    if dtype not in codes:
        raise TensorGameError("step_emit", -1, "dtype must be float32, float16 or bfloat16")
    dev = ring.device
    actions = _tokens(actions, (B,), S, dev, "actions")
    if out is None:
        out = torch.empty((B, T, S, S, S), dtype=dtype, device=dev)
    if out.dtype != dtype or tuple(out.shape) != (B, T, S, S, S) or not out.is_contiguous() or out.device != dev:
        raise TensorGameError("step_emit", -1, "out must be contiguous (B,T,S,S,S) of the requested dtype")
    if scalars is None:
        scalars = torch.empty((B, 1), dtype=torch.float32, device=dev)
    scalars = _flag(scalars, (B, 1), torch.float32, dev, "scalars")
    if done is None:
        done = torch.empty((B,), dtype=torch.uint8, device=dev)
    done = _flag(done, (B,), torch.uint8, dev, "done")
    overflow = _flag(overflow, (B,), torch.uint8, dev, "overflow")
    head = int(head_slot) % T
    with torch.cuda.device(dev):
        call("tg_step_emit", _ptr(ring), _ptr(actions), _ptr(out), _ptr(scalars), _ptr(done), _ptr(overflow), codes[dtype],
             B, S, T, head, C.c_float(float(t_step)), fs, gs, int(shift), _stream(dev))
    return out, scalars, done, (head + 1) % T


def state_hash(state) -> torch.Tensor:
    """64-bit key per game (int64 tensor holding the uint64 bits): the transposition-table key that
    replaces state_to_str (utils.py:164-169).  Equal states <=> equal keys (up to 2^-64 collisions)."""
    B, S, stride = _state_layout(state, "state")
    out = torch.empty((B,), dtype=torch.int64, device=state.device)
    with torch.cuda.device(state.device):
        call("tg_hash_u64", _ptr(state), _ptr(out), B, S, stride, _stream(state.device))
    return out


def alloc_seen_table(capacity: int, device) -> torch.Tensor:
    """An empty transposition table for ``seen``: int64 (capacity,) zeros, capacity a power of two (keep it at most
    half full)."""
    if capacity < 2 or capacity & (capacity - 1):
        raise TensorGameError("alloc_seen_table", -1, "capacity must be a power of two >= 2")
    return torch.zeros((capacity,), dtype=torch.int64, device=device)


def seen(keys, table, mask=None, insert: bool = False, status=None, fresh=None) -> torch.Tensor:
    """fresh[i] = mask[i] and keys[i] not in table (as it was BEFORE the call); with ``insert`` every masked key is in
    the table afterwards.  == the tree filter of extend_tree (act.py:188-195: ``c not in new_mc_tree``) and the
    recording of the expanded state's key (act.py:209-211), on the 64-bit keys of ``state_hash``.  ``keys`` may have any
    shape (int64 bits of the uint64 keys); ``mask`` uint8 of the same shape (e.g. ``changed`` of ``expand``);
    ``status`` int32 (1,): bit 0 set when the table was full.  Returns fresh (uint8, shape of keys)."""
    _need_gpu(keys, "keys")
    _need_gpu(table, "table")
    dev = keys.device
    if keys.dtype != torch.int64 or not keys.is_contiguous():
        raise TensorGameError("seen", -1, "keys must be contiguous int64 (the bits of the uint64 keys)")
    if table.dtype != torch.int64 or table.dim() != 1 or not table.is_contiguous() or table.device != dev:
        raise TensorGameError("seen", -1, f"table must be a contiguous int64 vector on {dev} (ops.alloc_seen_table)")
    n = keys.numel()
    if mask is not None:
        mask = _flag(mask, tuple(keys.shape), torch.uint8, dev, "mask")
    if fresh is None:
        fresh = torch.empty(tuple(keys.shape), dtype=torch.uint8, device=dev)
    fresh = _flag(fresh, tuple(keys.shape), torch.uint8, dev, "fresh")
    if status is not None and (status.dtype not in (torch.int32, torch.uint32) or status.numel() != 1 or status.device != dev):
        raise TensorGameError("seen", -1, f"status must be one 32-bit word on {dev}")
    with torch.cuda.device(dev):
        call("tg_seen_u64", _ptr(keys), _ptr(table), table.numel(), _ptr(fresh), _ptr(mask), _ptr(status), n,
             1 if insert else 0, _stream(dev))
    return fresh


def slice_rank(state) -> torch.Tensor:
    """int32 (B,): sum over the S slices state[b][i] of the exact rank of the S x S matrix
    (== get_rank per game, utils.py:134-140, which sums torch.linalg.matrix_rank over slices)."""
    B, S, stride = _state_layout(state, "state")
    out = torch.empty((B,), dtype=torch.int32, device=state.device)
    with torch.cuda.device(state.device):
        call("tg_rank_i32", _ptr(state), _ptr(out), B, S, stride, _stream(state.device))
    return out


def debug_fallbacks(device="cuda:0") -> int:
    """Workgroups that fell back from the packed 16-bit kernels to the exact byte-wise form so far
    (debug counter; synchronises the device)."""
    out = C.c_uint64(0)
    with torch.cuda.device(torch.device(device)):
        call("tg_debug_fallbacks", C.byref(out))
    return int(out.value)


def debug_handovers(device="cuda:0") -> int:
    """Games the matrix-core pass of step_many handed to the lattice kernels so far (debug counter; synchronises)."""
    out = C.c_uint64(0)
    with torch.cuda.device(torch.device(device)):
        call("tg_debug_handovers", C.byref(out))
    return int(out.value)
