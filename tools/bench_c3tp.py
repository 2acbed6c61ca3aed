"""conv3x3 forward: the halo-tile kernel against the tap-packed one (csrc/dense_fwd_tp.hip) at the encoder's three block
geometries, B = 64 (or argv[1]); band sizes swept.  HIP events, 10 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd import _lib
if os.environ.get("EML_LIB_PATH"):
    _lib.LIB_PATH = os.environ["EML_LIB_PATH"]   # experiment builds (tools/exp/c3tp_variants.sh)
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
GEOMS = [(240, 320, 120, 224), (120, 160, 204, 304), (60, 80, 246, 352)]
BANDS = (4, 6, 8, 10, 12, 15, 20, 24, 30, 40, 60)
if os.environ.get("C3TP_QUICK"):   # block 1 at the engine's band only
    GEOMS, BANDS = GEOMS[:1], (15, 30)
for (H, W, cin, ld) in GEOMS:
    P = B * H * W
    Z = torch.randn(P, 48, device="cuda"); X = torch.zeros(P, ld, device="cuda"); X2 = torch.zeros(P, ld, device="cuda")
    s2 = torch.rand(48, device="cuda") + 0.5; t2 = torch.randn(48, device="cuda") * 0.2
    W2 = torch.randn(12, 48, 3, 3, device="cuda") * 0.1
    W2p = torch.empty(6912, device="cuda"); W2t = torch.empty(5376, device="cuda")
    L.eml_dense_permute_w2_f32(p(W2), 12, p(W2p), st); L.eml_dense_permute_w2_tp_f32(p(W2), p(W2t), st)
    part = torch.zeros(1024 * 32, dtype=torch.float64, device="cuda")
    fl = 2.0 * P * 432 * 12
    ms = timeit(lambda: L.eml_dense_conv3x3_fwd_f32(p(Z), p(s2), p(t2), p(W2p), p(X), ld, cin, B, H, W, p(part), cu, st))
    print("%dx%d B=%d  halo-tile kernel: %.3f ms  %.1f TF/s (algorithmic)" % (H, W, B, ms, fl / ms / 1e9), flush=True)
    nw = L.eml_dense_conv3x3_fwd_tp_supported(B, H, W)
    for band in BANDS:
        if band > H: continue
        items = B * ((H + band - 1) // band)
        grid = min(items, 1024)
        ms = timeit(lambda: L.eml_dense_conv3x3_fwd_tp_f32(p(Z), p(s2), p(t2), p(W2t), p(X2), ld, cin, B, H, W, band, p(part), grid, st))
        print("   tap-packed nw=%d band=%3d items=%5d grid=%4d: %.3f ms  %.1f TF/s" % (nw, band, items, grid, ms, fl / ms / 1e9), flush=True)
    err = (X2[:, cin:cin + 12] - X[:, cin:cin + 12]).abs().max().item()
    print("   max |tp - halo| = %.3e (scale %.3e)" % (err, X[:, cin:cin + 12].abs().max().item()))
    del Z, X, X2
