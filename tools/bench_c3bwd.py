"""conv3x3 backward, fused data + weight gradient (eml_dense_conv3x3_bwd_fused_f32) at the encoder's block geometries;
EML_LIB_PATH selects an experiment build (tools/exp_build.sh nowtp -DEML_C3_WTP=0: round 4's weight gradient)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd import _lib
L, p, st = _lib.lib(), _lib.ptr, _lib.current_stream()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cu = torch.cuda.get_device_properties(0).multi_processor_count
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (H, W, cin, ld) in [(240, 320, 120, 224), (120, 160, 204, 304), (60, 80, 246, 352)]:
    P = B * H * W
    Gd = torch.randn(P, ld, device="cuda"); X = torch.randn(P, ld, device="cuda"); Z = torch.randn(P, 48, device="cuda")
    DZ = torch.empty(P, 48, device="cuda"); GF = torch.empty(P, 12, device="cuda")
    W2 = torch.randn(12, 48, 3, 3, device="cuda") * 0.1
    s2 = torch.rand(48, device="cuda") + 0.5; t2 = torch.randn(48, device="cuda") * 0.2
    zmean = torch.randn(48, device="cuda") * 0.1; zistd = torch.rand(48, device="cuda") + 0.5
    sB = torch.randn(ld, device="cuda") * 0.1; sC = torch.randn(ld, device="cuda") * 0.1
    part = torch.zeros(1024 * 96, dtype=torch.float64, device="cuda")
    partW = torch.empty(1024 * 2 * 27 * 256, device="cuda"); dW2 = torch.empty(12, 48, 3, 3, device="cuda")
    fl = 2 * 2.0 * P * 432 * 12
    ms = timeit(lambda: L.eml_dense_conv3x3_bwd_fused_f32(p(Gd), ld, cin, p(W2), p(Z), p(zmean), p(zistd), p(DZ), B, H, W, p(part), cu,
                                                          p(X), ld, cin, p(sB), p(sC), p(GF), p(s2), p(t2), p(partW), p(dW2), st))
    print("%dx%d B=%d  fused conv3x3 backward: %.3f ms  %.1f TF/s (algorithmic, data + weight gradient)" % (H, W, B, ms, fl / ms / 1e9), flush=True)
    del Gd, X, Z, DZ, GF
