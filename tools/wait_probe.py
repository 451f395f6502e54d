"""How long does a producer on a second stream need until its first release is visible to a resident stepper?
(The stepper's bounded wait must cover it: round 4 found stream creation + first launch on a new queue take > 1 s.)"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mat_mul_amd import ops
DEV = 'cuda:0'
for trial in range(3):
    for S, B in ((4, 300), (16, 70), (25, 26)):
        K = 4
        tok, tgt = ops.gen_demos(B, S, K, DEV, seed=1)
        st = ops.alloc_states(B, S, DEV); st.copy_(tgt)
        acts = tok.permute(1, 0, 2).contiguous()
        ready = torch.zeros(K, dtype=torch.int32, device=DEV)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        side, prod = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
        t1 = time.perf_counter()
        with torch.cuda.stream(side):
            ops.step_stream(st, acts, ready=ready, status=status)
        t2 = time.perf_counter()
        ts = []
        with torch.cuda.stream(prod):
            for k in range(K):
                ready[k:k + 1].fill_(1)
                prod.synchronize()
                ts.append(time.perf_counter() - t2)
        side.synchronize()
        t3 = time.perf_counter()
        print(trial, S, B, "streams %.3f s, stepper launch %.3f s, releases visible at" % (t1 - t0, t2 - t1),
              ["%.3f" % x for x in ts], "stepper done %.3f s, status %d" % (t3 - t2, int(status[0])), flush=True)
