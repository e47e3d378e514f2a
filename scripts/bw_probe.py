"""Device memory rates at the sizes of the box kernels (configs[2]'s IoU map is 93 MB written once): torch fill_ (write only), copy_ (read + write), sum (read only).
Round 5: what is the WRITE-ONLY rate the 8 TB/s peak should be compared with for iou_map_rows_kernel?"""
import torch
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (93, 128, 512, 2048):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    tf = t(lambda: x.fill_(1.0)); tz = t(lambda: x.zero_()); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.sum())
    print("%5d MB: fill %.2f TB/s (%.1f us), zero_ %.2f TB/s, copy %.2f TB/s (r+w), read(sum) %.2f TB/s" % (mb, mb / 1e6 / tf, tf * 1e6, mb / 1e6 / tz, 2 * mb / 1e6 / tc, mb / 1e6 / tr), flush=True)
