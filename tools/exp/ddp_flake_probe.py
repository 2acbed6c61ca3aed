"""Run-to-run reproducibility of two joint iterations (the worker of tests/test_gpu_ddp_two_ranks.py's one-rank RCCL test):
N plain runs and N one-rank-RCCL runs in fresh processes, every loss term and the parameter sum printed with the spread."""
import os, socket, sys, tempfile
import numpy as np
import torch.multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from test_gpu_ddp_two_ranks import _rccl_single_worker

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rows = {0: [], 1: []}
    for i in range(n):
        for dry in (0, 1):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            d = tempfile.mkdtemp()
            mp.spawn(_rccl_single_worker, args=(dry, port, d), nprocs=1, join=True)
            rows[dry].append(np.load(os.path.join(d, "s%d.npy" % dry)))
    np.set_printoptions(precision=9, linewidth=250)
    for dry in (0, 1):
        a = np.stack(rows[dry])[:, :-1]
        print("dry" if dry else "plain")
        print(a)
        print(" spread/|mean|:", (a.max(0) - a.min(0)) / np.abs(a.mean(0)))
    a = np.concatenate([np.stack(rows[0]), np.stack(rows[1])])[:, :-1]
    print("all runs spread/|mean|:", (a.max(0) - a.min(0)) / np.abs(a.mean(0)))
