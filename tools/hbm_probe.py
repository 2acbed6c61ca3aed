"""HBM ceilings of this part with plain torch kernels: fill (write only), copy (read + write), sum (read only)."""
import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.empty(1 << 30, device="cuda"); y = torch.empty_like(x)   # 4 GiB each
gb = x.numel() * 4 / 1e9
print("fill  %.2f TB/s" % (gb / t(lambda: x.fill_(1.0))))
print("copy  %.2f TB/s (read+write)" % (2 * gb / t(lambda: y.copy_(x)) ))
print("sum   %.2f TB/s" % (gb / t(lambda: x.sum())))
print("add   %.2f TB/s (2 reads + 1 write)" % (3 * gb / t(lambda: torch.add(x, y, out=y))))
