"""N4: packed demo files and the reference's on-disk layout.  CPU only."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from mat_mul_amd import demo_io
from oracle import tensor_game as O

REF = Path("/root/reference")


def _demos(B=5, S=4, R=6, seed=3):
    thr = O.categorical_thresholds((0.15, 0.7, 0.15))
    tok, tgt, _ = O.gen_demos_i8(B, S, R, thr, (-1, 0, 1), 1, seed=seed)
    return torch.from_numpy(tok), torch.from_numpy(tgt)


def test_packed_roundtrip(tmp_path):
    tok, tgt = _demos()
    f = tmp_path / "demos.tgd"
    demo_io.save_packed(f, tok, tgt, shift=1, seed=3, game_id_offset=10)
    assert f.stat().st_size == 64 + tok.numel() + tgt.numel()
    for mmap in (True, False):
        tok2, tgt2, meta = demo_io.load_packed(f, mmap=mmap)
        assert torch.equal(tok2, tok) and torch.equal(tgt2, tgt)
        assert meta == {"B": 5, "R": 6, "S": 4, "shift": 1, "seed": 3, "game_id_offset": 10}
    f.write_bytes(f.read_bytes()[:-1])
    with pytest.raises(ValueError, match="truncated"):
        demo_io.load_packed(f)


def test_reference_layout_roundtrip(tmp_path):
    tok, tgt = _demos(B=4, S=9, R=5)
    assert demo_io.export_reference_layout(tmp_path, tok, tgt) == 4
    seq = torch.load(tmp_path / "action_seq_2.pt")
    assert isinstance(seq, list) and len(seq) == 5 and seq[0].dtype == torch.int64 and seq[0].shape == (27,)
    t = torch.load(tmp_path / "target_tensor_2.pt")
    assert t.dtype == torch.float32 and t.shape == (9, 9, 9)
    tok2, tgt2 = demo_io.import_reference_layout(tmp_path, 4)
    assert torch.equal(tok2, tok) and torch.equal(tgt2, tgt)


@pytest.mark.skipif(not REF.exists(), reason="the reference only exists in the build container")
def test_reference_dataset_reads_our_files(tmp_path):
    """The reference's own SyntheticDemoDataset(overwrite=False) consumes the exported files, and its
    __getitem__ (datasets.py:78-122) returns what the oracle predicts for our tokens."""
    tok, tgt = _demos(B=3, S=4, R=7, seed=11)
    demo_io.export_reference_layout(tmp_path / "d", tok, tgt)
    np.save(tmp_path / "tok.npy", tok.numpy())
    np.save(tmp_path / "tgt.npy", tgt.numpy())
    script = f"""
import sys, numpy as np, torch
sys.dont_write_bytecode = True
sys.path.insert(0, {str(REF)!r})
import datasets
ds = datasets.SyntheticDemoDataset(7, 3, 2, 4, "cpu", overwrite=False, save_dir={str(tmp_path / 'd')!r})
assert ds.n_demos == 3 and len(ds) == 21
out = [ds[i] for i in range(21)]
np.save({str(tmp_path / 'frames.npy')!r}, np.stack([o[0].numpy() for o in out]))
np.save({str(tmp_path / 'meta.npy')!r}, np.array([[o[1].item(), o[3].item()] for o in out]))
np.save({str(tmp_path / 'act.npy')!r}, np.stack([o[2].numpy() for o in out]))
"""
    subprocess.run([sys.executable, "-c", script], check=True, cwd=tmp_path, timeout=300)
    frames, meta, act = (np.load(tmp_path / n) for n in ("frames.npy", "meta.npy", "act.npy"))
    for i in range(21):
        d, ia = divmod(i, 7)
        f, sc, a, rw = O.demo_getitem(list(tok[d].numpy()), tgt[d].numpy(), ia, 2)
        assert np.array_equal(frames[i], f.astype(np.float32)) and meta[i].tolist() == [sc, rw]
        assert np.array_equal(act[i], a)
