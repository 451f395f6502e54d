"""On-disk formats for synthetic demonstrations.

The reference stores TWO pickles per demo -- ``action_seq_{i}.pt`` (a Python list of R int64
tensors of shape (3S,)) and ``target_tensor_{i}.pt`` (float32 (S,S,S)) -- and ``torch.load``s a
pair per sample (datasets.py:62-69, 86-89).  Here a whole dataset is ONE packed int8 file that
can be memory-mapped; ``export_reference_layout`` / ``import_reference_layout`` convert to and
from the reference's layout, so demos generated on the GPU can be read by the reference's own
``SyntheticDemoDataset(overwrite=False, save_dir=...)``.

Packed file: 64-byte header ``TGDEMOS1`` + little-endian int64 (B, R, S, shift, seed,
game_id_offset, 0) then tokens int8 (B,R,3S) then targets int8 (B,S,S,S).
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Dict, Tuple

import numpy as np
import torch

MAGIC = b"TGDEMOS1"
HEADER = struct.Struct("<8s7q")
assert HEADER.size == 64


def save_packed(path, tokens: torch.Tensor, targets: torch.Tensor, shift: int = 1, seed: int = 0,
                game_id_offset: int = 0) -> None:
    """tokens int8 (B,R,3S), targets int8 (B,S,S,S) (any device) -> one packed file."""
    B, R, S3 = tokens.shape
    S = S3 // 3
    if tokens.dtype != torch.int8 or targets.dtype != torch.int8 or tuple(targets.shape) != (B, S, S, S):
        raise ValueError("tokens must be int8 (B,R,3S) and targets int8 (B,S,S,S)")
    with open(path, "wb") as f:
        f.write(HEADER.pack(MAGIC, B, R, S, shift, seed & (2 ** 63 - 1), game_id_offset, 0))
        f.write(tokens.detach().cpu().contiguous().numpy().tobytes())
        f.write(targets.detach().cpu().contiguous().numpy().tobytes())


def load_packed(path, device=None, mmap: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, int]]:
    """-> (tokens int8 (B,R,3S), targets int8 (B,S,S,S), meta).  With ``mmap`` the host arrays are
    memory-mapped (nothing is read until touched); ``device`` uploads them."""
    with open(path, "rb") as f:
        magic, B, R, S, shift, seed, gid0, _ = HEADER.unpack(f.read(HEADER.size))
    if magic != MAGIC:
        raise ValueError(f"{path}: not a packed demo file")
    n_tok, n_tgt = B * R * 3 * S, B * S ** 3
    if Path(path).stat().st_size != HEADER.size + n_tok + n_tgt:
        raise ValueError(f"{path}: truncated")
    if mmap:
        raw = np.memmap(path, dtype=np.int8, mode="r", offset=HEADER.size, shape=(n_tok + n_tgt,))
    else:
        raw = np.fromfile(path, dtype=np.int8, offset=HEADER.size)
    tok = torch.from_numpy(np.array(raw[:n_tok]).reshape(B, R, 3 * S))
    tgt = torch.from_numpy(np.array(raw[n_tok:]).reshape(B, S, S, S))
    if device is not None:
        tok, tgt = tok.to(device), tgt.to(device)
    return tok, tgt, {"B": B, "R": R, "S": S, "shift": shift, "seed": seed, "game_id_offset": gid0}


def export_reference_layout(save_dir, tokens: torch.Tensor, targets: torch.Tensor, start_index: int = 0) -> int:
    """Write the reference's per-demo files (datasets.py:62-69): ``action_seq_{i}.pt`` = list of R
    int64 (3S,) tensors, ``target_tensor_{i}.pt`` = float32 (S,S,S).  Returns the number written."""
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    tok = tokens.detach().cpu().to(torch.int64)
    tgt = targets.detach().cpu().to(torch.float32)
    for d in range(tok.shape[0]):
        torch.save([tok[d, r].clone() for r in range(tok.shape[1])], save_dir / f"action_seq_{start_index + d}.pt")
        torch.save(tgt[d].clone(), save_dir / f"target_tensor_{start_index + d}.pt")
    return tok.shape[0]


def import_reference_layout(save_dir, n_demos: int, start_index: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Read ``n_demos`` demos the reference wrote (datasets.py:86-89) into packed int8 arrays."""
    save_dir = Path(save_dir)
    toks, tgts = [], []
    for d in range(start_index, start_index + n_demos):
        seq = torch.load(save_dir / f"action_seq_{d}.pt")
        tgt = torch.load(save_dir / f"target_tensor_{d}.pt")
        t = torch.stack(list(seq))
        if int(t.min()) < -128 or int(t.max()) > 127 or float(tgt.abs().max()) > 127 or not torch.equal(tgt, tgt.round()):
            raise ValueError(f"demo {d} does not fit int8")
        toks.append(t.to(torch.int8))
        tgts.append(tgt.to(torch.int8))
    return torch.stack(toks), torch.stack(tgts)
