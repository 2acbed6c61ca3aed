import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for B in (64, 256):
    r = bench.time_sinkhorn(B, 128, .05, "cuda:0", reps=50)
    print(B, r["ms_per_loss_call"], r["sweeps"], r["ms_per_eps_step"], r["frac_of_hbm_peak_8TBps"])
