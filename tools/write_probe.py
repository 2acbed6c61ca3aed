import torch
def t(fn,n=20):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for mb in (128, 537, 2048):
    x=torch.empty(mb*1024*1024//4, device='cuda'); y=torch.empty_like(x)
    f=t(lambda: x.fill_(1.0)); c=t(lambda: y.copy_(x)); m=t(lambda: x.mul_(2.0))
    print("%5d MB: fill %.1f us = %.2f TB/s written | copy %.1f us = %.2f TB/s (r+w) | x*=2 %.1f us = %.2f TB/s (r+w)" % (mb, f*1e3, mb*1.048576e6/f/1e9, c*1e3, 2*mb*1.048576e6/c/1e9, m*1e3, 2*mb*1.048576e6/m/1e9))
