#!/usr/bin/env python3
"""Game strides of 16-byte multiples (15 632 B at S=25, 736 B at S=9) against 128-byte multiples (15 744 B, 768 B):
tg_step_i8 (in place), tg_expand_i8 (k = 8) and tg_emit_frames (T = 4) by hipGraph replay (bench.py's machinery).
    python tools/stride_ab.py [S B ...]            (default: 25 4096 25 32768 9 32768 9 262144)
SURVEY.md section 7, hard part 4: "use a game_stride_bytes parameter (e.g. 15 744 = 123 x 128) and test both"."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from mat_mul_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
args = [int(x) for x in sys.argv[1:]] or [25, 4096, 25, 32768, 9, 32768, 9, 262144]
for S, B in zip(args[0::2], args[1::2]):
    N = S ** 3
    for pad in (16, 128):
        big = B * N > (64 << 20)
        # ---- step, in place ----
        st, sc, _ = bench.make_demo_schedule(B, S, 8, dev, 1, 0, pad_to=pad)
        tm = bench.StepTimer(st, sc, dev, "graph", pad_to=pad)
        K = 32 if big else 256
        r = tm.measure(K, 16, 5)
        ro = bench.roofline(B, S, K, r["event_ms"], bench.needed_bytes_per_launch(B, S, sc))
        print(f"S={S} B={B} stride={st.stride(0)}: step   ok={r['ok']} {ro['avg_launch_us']:8.2f} us  frac={ro['frac']}", flush=True)
        del tm
        # ---- expand, k = 8 ----
        k = 8
        tok = torch.stack(sc[:k], dim=1).contiguous()
        kids = ops.alloc_states(B * k, S, dev, pad_to=pad).unflatten(0, (B, k))
        kd = torch.zeros((B, k), dtype=torch.uint8, device=dev)
        kc = torch.zeros((B, k), dtype=torch.uint8, device=dev)
        sec = bench.graph_time(lambda: ops.expand(st, tok, out=kids, done=kd, changed=kc), dev, reps=5 if big else 20)
        one, dn = ops.step(st, tok[:, k - 1].contiguous())
        ok = bool(torch.equal(kids[:, k - 1], one)) and bool(torch.equal(kd[:, k - 1], dn))
        nb = B * (N + k * (N + 3 * S + 2))
        print(f"S={S} B={B} stride={st.stride(0)}: expand ok={ok} {sec * 1e6:8.2f} us  frac={nb / sec / 8e12:.3f}", flush=True)
        del kids, kd, kc, one
        # ---- emit_frames, T = 4 ----
        T = 4
        ring = ops.alloc_ring(B, S, T, dev, pad_to=pad)
        for f in range(T):
            ring[:, f].copy_(st)
        scal = torch.empty((B, 1), dtype=torch.float32, device=dev)
        for dt, w in [(torch.float32, 4), (torch.float16, 2)]:
            x = torch.empty((B, T, S, S, S), dtype=dt, device=dev)
            sec = bench.graph_time(lambda: ops.emit_frames(ring, 1, 1.0, dt, out=x, scalars=scal), dev, reps=5 if big else 20)
            ok = bool(torch.equal(x[:, 0].to(torch.int8), st))
            print(f"S={S} B={B} frame stride={ring.stride(1)}: emit {str(dt)[6:]:8s} ok={ok} {sec * 1e6:8.2f} us  "
                  f"frac={B * (T * N * (1 + w) + 4) / sec / 8e12:.3f}", flush=True)
            del x
        del ring, st, sc, tok
        torch.cuda.empty_cache()
