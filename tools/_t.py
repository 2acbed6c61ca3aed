import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from emlight_amd.GenProjector.spherenet import SphereConv2D
SphereConv2D.fused_min_bytes = 0
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for C, O, H, W in [(128, 256, 128, 256), (128, 128, 128, 256), (128, 64, 128, 256), (64, 64, 128, 256), (128, 512, 64, 128), (256, 128, 64, 128), (512, 256, 32, 64)]:
    B = 32
    m = SphereConv2D(C, O).cuda()
    x = torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    gf = 2.0 * B * H * W * 9 * C * O / 1e9
    with torch.no_grad():
        tf = ev(lambda: m(x))
    m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    xg = x.clone().requires_grad_(True); y = m(xg); gy = torch.randn_like(y)
    td = ev(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
    m.weight.requires_grad_(True)
    y = m(x); 
    tw = ev(lambda: torch.autograd.grad(y, m.weight, gy, retain_graph=True))
    print("C=%4d O=%4d %3dx%3d fwd %.3f ms %.1f TF/s | dgrad %.3f ms %.1f | wgrad %.3f ms %.1f" % (C, O, H, W, tf, gf / tf, td, gf / td, tw, gf / tw), flush=True)
