"""Host-side enqueue time of one regression step vs its GPU time (is the step launch-bound at small batch?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emlight_amd.RegressionNetwork.engine import RegressionTrainer
from emlight_amd.RegressionNetwork.data import synthetic_batch
for B in (64, 32, 8):
    tr = RegressionTrainer(anchors=128, crop_hw=(240, 320), blur=.05, device="cuda:0", world=1)
    batch = synthetic_batch(B, 128, (240, 320), seed=1, device="cuda:0")
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize()
    host, t0 = 0.0, time.perf_counter()
    for _ in range(5):
        h0 = time.perf_counter()
        tr.step(batch)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("B=%d  step %.1f ms  host enqueue %.1f ms per step" % (B, tot / 5 * 1e3, host / 5 * 1e3))
    del tr
