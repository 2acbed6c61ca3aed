"""Which orientation of the SphereConv weight-gradient GEMM does rocBLAS/hipBLASLt run faster?"""

import torch
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rows, C, O in [(32 * 32768, 128, 256), (32 * 32768, 128, 64), (32 * 8192, 256, 512), (32 * 2048, 512, 1024), (32 * 512, 1024, 1024)]:
    a9 = torch.randn(rows, 9 * C, device="cuda"); gy = torch.randn(rows, O, device="cuda")
    fl = 2.0 * rows * 9 * C * O
    m1 = t(lambda: gy.t() @ a9)
    m2 = t(lambda: (a9.t() @ gy))
    w2 = torch.randn(O, 9 * C, device="cuda")
    m3 = t(lambda: a9 @ w2.t())
    m4 = t(lambda: gy @ w2)
    print("rows %8d C %4d O %4d | wgrad gy^T@a9 %.2f ms %.0f TF | a9^T@gy %.2f ms %.0f TF | fwd %.2f ms %.0f TF | dA9 %.2f ms %.0f TF"
          % (rows, C, O, m1, fl / m1 / 1e9, m2, fl / m2 / 1e9, m3, fl / m3 / 1e9, m4, fl / m4 / 1e9))
    del a9, gy, w2
