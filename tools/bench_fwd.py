import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd.RegressionNetwork.DenseNet import DenseNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = "hip"
hw = (240, 320)
net = DenseNet(anchors=128, crop_hw=hw).cuda().train()
x = torch.rand(B, 3, *hw, device="cuda")
with torch.no_grad():
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
flops = 43.775e9 * B
print("engine %s B=%d fwd %.2f ms  %.1f img/s  %.1f TFLOP/s (%.1f%% of 157.3)" % (eng, B, dt * 1e3, B / dt, flops / dt / 1e12, flops / dt / 157.3e10))
