"""The reference's env step in the REFERENCE'S OWN dtypes and op sequence, on torch-CPU.
TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py).

This is what ``bench.py``'s ``cpu_baseline`` leg times (kind "port"): float32 ``(B,T,S,S,S)``
state, int64 ``(B,k,3S)`` tokens, the same broadcast outer product -> promote-subtract ->
history ``cat`` as ``get_child_states`` (reference act.py:266-275, utils.py:56-111), plus the
batched form of ``tensor_factorized`` (utils.py:181-188).  It is checked against the numpy
oracle and the golden fixtures in tests/test_oracle_golden.py::test_ref_dtype_port.
"""
from __future__ import annotations

import torch


def action_to_tensor(action: torch.Tensor, shift: int = 1) -> torch.Tensor:
    """utils.py:56-96: split after the shift, then the unsqueeze-broadcast outer product."""
    dim_3d = action.shape[-1] // 3
    uu, vv, ww = (action - shift).split(dim_3d, dim=-1)
    return uu.unsqueeze(-1).unsqueeze(-1) * vv.unsqueeze(-1).unsqueeze(-3) * ww.unsqueeze(-2).unsqueeze(-3)


def get_child_states(state: torch.Tensor, actions: torch.Tensor, shift: int = 1):
    """act.py:266-275, line for line in meaning: float32 head minus int64 action tensor
    (type promotion to float32), then one ``cat`` per candidate."""
    k = actions.shape[1]
    action_tensor = action_to_tensor(actions, shift)
    initial_head_state = state[:, 0].unsqueeze(1)
    new_head_states = initial_head_state - action_tensor
    return [torch.cat([new_head_states[:, i:i + 1], state[:, :-1]], dim=1) for i in range(k)]


def done_per_game(state: torch.Tensor) -> torch.Tensor:
    """utils.py:181-188 applied to every game's head: ``(head == 0).all()`` per game."""
    return (state[:, 0] == 0).flatten(1).all(dim=1)


def env_step(state: torch.Tensor, actions: torch.Tensor, shift: int = 1):
    """One batched env step the way the reference's functions compose (k = 1)."""
    new_state = get_child_states(state, actions, shift)[0]
    return new_state, done_per_game(new_state)
