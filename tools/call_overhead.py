import sys, time
sys.path.insert(0, '.')
import torch
from mat_mul_amd import ops, TensorGameEnv
for B, S in [(1, 4), (16, 4), (1, 16)]:
    env = TensorGameEnv(B, S, "cuda:0")
    env.reset()
    a = torch.ones((B, 3 * S), dtype=torch.int8, device="cuda:0")
    k8 = torch.ones((B, 8, 3 * S), dtype=torch.int8, device="cuda:0")
    for name, fn in [("env.step", lambda: env.step(a)), ("ops.expand k=8", lambda: ops.expand(env.state, k8)),
                     ("ops.state_hash", lambda: ops.state_hash(env.state))]:
        for _ in range(50): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(2000): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2000
        print(f"B={B} S={S} {name}: {dt*1e6:.1f} us per call (host+device, back to back)")
