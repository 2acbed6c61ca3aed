"""Time individual dense-engine kernels at training shapes (B=64, 240x320) with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd import _lib
if os.environ.get('EML_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EML_LIB_PATH']   # experiment builds (tools/exp_build.sh)
L, p = _lib.lib(), _lib.ptr
dev = "cuda"
B = 64
sel = sys.argv[1:]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
st = _lib.current_stream()
G = int(os.environ.get("EML_G", "512"))
for (Hb, Wb, Cin, ld) in [(240, 320, 48, 224), (240, 320, 120, 224), (240, 320, 204, 224), (120, 160, 204, 304), (60, 80, 246, 352)]:
    P = B * Hb * Wb
    Kp = (Cin + 15) // 16 * 16
    X = torch.randn(P, ld, device=dev)
    Gd = torch.randn(P, ld, device=dev)
    Z = torch.randn(P, 48, device=dev); DZ = torch.randn(P, 48, device=dev)
    s1 = torch.rand(Kp, device=dev) + 0.5; t1 = torch.randn(Kp, device=dev) * 0.2
    s2 = torch.rand(48, device=dev) + 0.5; t2 = torch.randn(48, device=dev) * 0.2
    mean = torch.zeros(ld, device=dev); istd = torch.ones(ld, device=dev)
    cA = torch.rand(48, device=dev); cB = torch.rand(48, device=dev) * 0.1; cC = torch.rand(48, device=dev) * 0.1
    W1 = torch.randn(48, Cin, device=dev) * 0.1; W2 = torch.randn(12, 48, 3, 3, device=dev) * 0.1
    W1p = torch.empty(Kp * 48, device=dev); W2p = torch.empty(6912, device=dev); Wd = torch.empty(Kp * 48, device=dev)
    L.eml_dense_permute_w1_f32(p(W1), 48, Cin, Kp, p(W1p), st); L.eml_dense_permute_w2_f32(p(W2), 12, p(W2p), st)
    L.eml_dense_permute_w1_bwd_f32(p(W1), 48, Cin, Kp, 48, p(Wd), st)
    GF = torch.empty(P, 12, device=dev)
    part = torch.zeros(4 * 1024 * 96, dtype=torch.float64, device=dev)
    partW = torch.empty(1024 * 352 * 48, device=dev)
    dW1 = torch.empty(48, Cin, device=dev); dW2 = torch.empty(12, 48, 3, 3, device=dev)
    fl1 = 2.0 * P * Cin * 48; fl2 = 2.0 * P * 432 * 12
    fns = {
        "c1x1_fwd": (lambda: L.eml_dense_conv1x1_fwd_f32(p(X), ld, P, Hb, Wb, 0, Kp, p(s1), p(t1), p(W1p), 48, p(Z), 48, p(part), G, st), fl1),
        "c3x3_fwd": (lambda: L.eml_dense_conv3x3_fwd_f32(p(Z), p(s2), p(t2), p(W2p), p(X), ld, Cin, B, Hb, Wb, p(part), G, st), fl2),
        "c3x3_bwd_data": (lambda: L.eml_dense_conv3x3_bwd_data_f32(p(Gd), ld, Cin, p(W2), p(Z), p(mean), p(istd), p(DZ), B, Hb, Wb, p(part), G, p(X), ld, p(mean), p(istd), p(GF), st), fl2),
        "c3x3_bwd_wgt": (lambda: L.eml_dense_conv3x3_bwd_weight_f32(p(Gd), ld, Cin, p(Z), p(s2), p(t2), B, Hb, Wb, p(partW), p(dW2), G, st), fl2),
        "c1x1_bwd_wgt": (lambda: L.eml_dense_conv1x1_bwd_weight_f32(p(X), ld, P, Hb, Wb, 0, Kp, Cin, p(s1), p(t1), p(DZ), 48, p(Z), 48, p(cA), p(cB), p(cC), 48, p(partW), p(dW1), G, st), fl1),
        "c1x1_bwd_data": (lambda: L.eml_dense_conv1x1_bwd_data_f32(p(DZ), 48, p(Z), 48, p(cA), p(cB), p(cC), 48, p(Wd), p(X), ld, p(s1), p(t1), p(mean), p(istd), P, Hb, Wb, 0, Kp, p(Gd), ld, 1, p(part), G, st), fl1),
    }
    print("== %dx%d Cin=%d ld=%d P=%.2fM" % (Hb, Wb, Cin, ld, P / 1e6))
    for k, (fn, fl) in fns.items():
        if sel and k not in sel:
            continue
        ms = timeit(fn)
        print("   %-15s %8.3f ms  %6.1f TFLOP/s" % (k, ms, fl / ms / 1e9))
    del X, Gd, Z, DZ
