import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (128, 512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.sum())
    print("%5d MB: fill %.2f TB/s, copy %.2f TB/s (r+w), read(sum) %.2f TB/s" % (mb, mb / 1e6 * 1.048576 / tf, 2 * mb / 1e6 * 1.048576 / tc, mb / 1e6 * 1.048576 / tr))
