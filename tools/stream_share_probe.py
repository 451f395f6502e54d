"""tg_step_stream_i8 at S=4: per-step time with ready words (pre-set) against no ready words, with and without progress.
VERDICT r3 item 2c: 6.14 ps per game-step at 131 072 games (resident, ready + progress) against 4.98 at 2^20 in rounds."""
import statistics, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mat_mul_amd import ops
dev = torch.device('cuda:0')

def run(b2, k2, use_ready, use_prog, s2=4, r2=7, reps=7):
    tok, tgt = ops.gen_demos(b2, s2, r2, dev, seed=4)
    cyc = torch.cat([tok, tok], dim=1)
    cyc[:, r2:, :s2] = 2 - cyc[:, r2:, :s2]
    acts = cyc.permute(1, 0, 2).contiguous().repeat(k2 // (2 * r2), 1, 1)
    st = ops.alloc_states(b2, s2, dev); st.copy_(tgt)
    dn = torch.empty((k2, b2), dtype=torch.uint8, device=dev)
    cap = ops.step_stream_capacity(s2, dev)
    n_units = ops.step_stream_layout(b2, s2, dev)[0] if b2 <= cap else -(-b2 // 64)
    ready = torch.ones(k2, dtype=torch.int32, device=dev) if use_ready else None
    prog = torch.zeros(n_units, dtype=torch.int32, device=dev) if use_prog else None
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    fn = lambda: ops.step_stream(st, acts, done=dn, ready=ready, progress=prog, status=status)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    sec = statistics.median(ts)
    ok = bool(torch.equal(st, tgt)) and int(status[0]) == 0
    print(f"B={b2:8d} K={k2:5d} ready={use_ready!s:5} progress={use_prog!s:5}: {sec / k2 * 1e6:7.3f} us/step  "
          f"{sec / k2 / b2 * 1e12:6.2f} ps/game-step  ok={ok}", flush=True)

if __name__ == "__main__":
    import os
    print("TG_LIB_VARIANT=%s TG_STREAM_NO_STAGGER=%s" % (os.environ.get("TG_LIB_VARIANT"), os.environ.get("TG_STREAM_NO_STAGGER")))
    sizes = ((65536, 1008), (131072, 504), (262144, 252), (1 << 20, 112))
    if "small" in sys.argv:
        sizes = ((1024, 1008), (4096, 1008), (8192, 1008), (16384, 1008), (24576, 1008), (32768, 1008), (49152, 1008), (65536, 1008), (98304, 504))
    for b2, k2 in sizes:
        cap = ops.step_stream_capacity(4, dev)
        for use_ready in ((True, False) if b2 <= cap else (False,)):
            for use_prog in ((True,) if "small" in sys.argv else (True, False)):
                run(b2, k2, use_ready, use_prog)
