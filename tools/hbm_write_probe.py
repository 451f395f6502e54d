import torch, time
x = torch.empty(512*1024*1024, dtype=torch.int8, device="cuda")
y = torch.empty_like(x)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e-3/it
s = t(lambda: x.fill_(3)); print("fill 512MB: %.1f us  %.2f TB/s" % (s*1e6, x.numel()/s/1e12))
s = t(lambda: y.copy_(x)); print("copy 512MB: %.1f us  %.2f TB/s (r+w)" % (s*1e6, 2*x.numel()/s/1e12))
xi = x.view(torch.int32)
s = t(lambda: xi.fill_(3)); print("fill int32: %.1f us  %.2f TB/s" % (s*1e6, x.numel()/s/1e12))
