"""The reference's hot-path functions, same names and argument meaning, on the GPU.

Each function states the reference ``file:line`` it replaces.  States are int8 ROCm tensors
(the reference uses float32 holding small integers), tokens int8 (reference: int64).
Errors: ``TensorGameError`` for anything the kernels refuse (the reference raises torch shape
errors and never range-checks tokens, utils.py:64-66).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import ops
from ._lib import TensorGameError


def _as_state(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int8 else t.to(torch.int8)


def action_to_uvw(action: torch.Tensor, shift: int = 1):
    """reference utils.py:56-66: ``(action - shift).split(dim_3d, dim=-1)`` (views; int16 so the
    subtraction cannot wrap)."""
    dim_3d = action.shape[-1] // 3
    return (action.to(torch.int16) - shift).split(dim_3d, dim=-1)


def action_to_tensor(action: torch.Tensor, shift: int = 1) -> torch.Tensor:
    """reference utils.py:88-96 (and uvw_to_tensor :69-85): tokens (*,3S) -> int8 (*,S,S,S)."""
    S = action.shape[-1] // 3
    lead = action.shape[:-1]
    tok = ops.as_tokens(action).reshape(-1, 1, 3 * S)
    out = ops.gen_from_factors(tok, S, shift=shift)
    return out.reshape(*lead, S, S, S)


def get_head_state(state: torch.Tensor, unsqueeze: bool = True) -> torch.Tensor:
    """reference utils.py:99-111."""
    return state[:, 0].unsqueeze(1) if unsqueeze else state[:, 0]


def tensor_factorized(state: torch.Tensor) -> torch.Tensor:
    """reference utils.py:181-188, VERBATIM semantics ``(state[0] == 0).all()``: index 0 of the
    leading axis only.  For the per-game terminal flags of a batch use ``ops.done``."""
    s0 = _as_state(state)[0]
    S = s0.shape[-1]
    return ops.done(s0.reshape(-1, S, S, S).contiguous()).all()


def get_child_states(state: torch.Tensor, actions: torch.Tensor, vec_cardinality: int = 5, shift: int = 1) -> List[torch.Tensor]:
    """reference act.py:266-275: state (B,T,S,S,S), actions (B,k,3S) -> list of k (B,T,S,S,S);
    new head = head - action tensor, history shifts right (oldest frame dropped)."""
    state = _as_state(state)
    B, T, S = state.shape[0], state.shape[1], state.shape[2]
    k = actions.shape[1]
    head = state[:, 0].contiguous()
    kids, _, _ = ops.expand(head, ops.as_tokens(actions, state.device), shift=shift)
    if T == 1:
        return [kids[:, i:i + 1] for i in range(k)]
    hist = state[:, :-1]
    return [torch.cat([kids[:, i:i + 1], hist], dim=1) for i in range(k)]


def remove_null_actions(state: torch.Tensor, candidate_states: Sequence[torch.Tensor]) -> List[int]:
    """reference utils.py:191-194: indexes of candidates whose head differs from the parent's
    anywhere in the batch."""
    head = _as_state(state)[:, 0]
    return [i for i, c in enumerate(candidate_states) if bool((_as_state(c)[:, 0] != head).any())]


def state_to_key(state: torch.Tensor) -> torch.Tensor:
    """The dict key of the search tree: reference utils.py:164-169 ``state_to_str(get_head_state(state))`` builds a
    127-character string per state; here a 64-bit key per game (int64 bits), equal exactly when the head frames are
    equal (up to 2^-64 collisions).  state (B,T,S,S,S) or (B,S,S,S)."""
    s = _as_state(state)
    head = s[:, 0] if s.dim() == 5 else s
    S = head.shape[-1]
    return ops.state_hash(head if head.stride()[1:] == (S * S, S, 1) else head.contiguous())


def expand_new_candidates(state: torch.Tensor, actions: torch.Tensor, tree, shift: int = 1):
    """The candidate filter at a leaf of reference ``extend_tree`` (act.py:183-195) for a BATCH of leaves: children =
    ``get_child_states`` (:183), drop null actions (``remove_null_actions``, :185), drop children whose key is already
    in the tree (``c not in new_mc_tree``, :188-195).  ``tree`` is a ``TranspositionTable``.  state (B,T,S,S,S) or
    (B,S,S,S) int8, actions (B,k,3S).  Returns (children int8 (B,k,S,S,S), keep uint8 (B,k), keys int64 (B,k),
    done uint8 (B,k)); the caller records an expanded leaf with ``tree.insert(state_to_key(state))`` (act.py:209-211)."""
    s = _as_state(state)
    head = (s[:, 0] if s.dim() == 5 else s).contiguous()
    kids, done, changed, keys = ops.expand(head, ops.as_tokens(actions, head.device), shift=shift, want_keys=True)
    return kids, tree.fresh(keys, mask=changed), keys, done


def take_actions(action_seq, target_tensor: torch.Tensor, shift: int = 1) -> torch.Tensor:
    """reference datasets.py:144-153 (_take_actions): target (S,S,S) minus every action of the list."""
    target = _as_state(target_tensor)
    S = target.shape[-1]
    seq = action_seq if isinstance(action_seq, torch.Tensor) else (
        torch.stack(list(action_seq)) if len(action_seq) else torch.empty((0, 3 * S), dtype=torch.int8))
    if seq.shape[0] == 0:
        return target.clone()
    tok = ops.as_tokens(seq, target.device).reshape(1, -1, 3 * S)
    out, _ = ops.step_many(target.reshape(1, S, S, S).contiguous(), tok, shift=shift)
    return out.reshape(target.shape)


def build_matmul_tensor(dim_t: int, dim_i: int, dim_j: int, dim_k: int, device="cuda") -> torch.Tensor:
    """reference utils.py:143-161: <n,n,n> in frame 0 of (dim_t,S,S,S), zeros elsewhere.  Square
    shapes only (the reference's index formula is wrong otherwise)."""
    if not (dim_i == dim_j == dim_k):
        raise TensorGameError("build_matmul_tensor", -1, "only square shapes are defined")
    S = dim_i * dim_i
    out = torch.zeros((dim_t, S, S, S), dtype=torch.int8, device=device)
    ops.reset_matmul(out[:1], dim_i)
    return out


def create_synthetic_demo(values, probs, n_actions: int, dim_3d: int, shift: int, seed: int = 0,
                          device="cuda") -> Tuple[List[torch.Tensor], torch.Tensor]:
    """reference utils.py:203-233: (list of n_actions token vectors (3S,), target (S,S,S))."""
    vals = [int(v) for v in (values.tolist() if hasattr(values, "tolist") else values)]
    pr = [float(p) for p in (probs.tolist() if hasattr(probs, "tolist") else probs)]
    actions, target = ops.gen_demos(1, dim_3d, n_actions, device, values=vals, probs=pr, shift=shift, seed=seed)
    return list(actions[0]), target[0]


def uvw_to_demo(uu: torch.Tensor, vv: torch.Tensor, ww: torch.Tensor, device="cuda", shift: int = 1):
    """reference utils.py:40-53: (sum_i u_i(x)v_i(x)w_i, token table cat(uu,vv,ww)+shift)."""
    tokens = ops.as_tokens(torch.cat((uu, vv, ww), dim=1) + shift, device)
    S = uu.shape[1]
    return ops.gen_from_factors(tokens.unsqueeze(0), S, shift=shift)[0], tokens


def take_action(state_batch: torch.Tensor, tokens: torch.Tensor, n_samples: int, shift: int = 2):
    """reference training.py:249-268 (SyntheticDemoTrainingApp._take_action, minus the model call that
    produces ``tokens``): one batched env step with history shift, the per-game non-zero count
    (``rank_ubs``, :266) grouped by ``n_samples``, and the best sample per group (:267).
    state_batch int8 (B,T,S,S,S); tokens (B,3S) with the reference's ``- 2`` shift (:253).
    Returns (new_state_batch (B,T,S,S,S), rank_ubs int32 (B/n_samples, n_samples), best (values, indices))."""
    state = _as_state(state_batch)
    B, T, S = state.shape[0], state.shape[1], state.shape[2]
    head = state[:, 0].contiguous()
    new_head, _ = ops.step(head, ops.as_tokens(tokens, state.device), shift=shift)
    new_state = torch.cat((new_head.unsqueeze(1), state), dim=1)[:, :-1]   # :256-258
    _, nnz = ops.done(new_head, want_nnz=True)
    rank_ubs = nnz.view(-1, n_samples)
    return new_state, rank_ubs, torch.min(rank_ubs, -1)
