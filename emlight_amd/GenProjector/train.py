"""Entry point mirroring ``GenProjector/train.py``: per iteration one generator step and one discriminator step
(``train.py:28-37``), torchrun-aware.  ``--synthetic`` batches replace the licence-restricted Laval dataset.

    python -m emlight_amd.GenProjector.train --synthetic --batchSize 8 --max_iters 10
    torchrun --nproc-per-node 8 -m emlight_amd.GenProjector.train --synthetic
"""
import argparse
import os

import torch

from ..RegressionNetwork.engine import init_distributed
from . import data, networks
from .model_trainer import Trainer


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="laval")
    ap.add_argument("--checkpoints_dir", default="./checkpoints")
    ap.add_argument("--batchSize", type=int, default=16, help="per-GPU batch (reference: 16 over 2 GPUs)")
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--ndf", type=int, default=64)
    ap.add_argument("--niter", type=int, default=50)
    ap.add_argument("--niter_decay", type=int, default=0)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--no_TTUR", action="store_true")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--iters_per_epoch", type=int, default=100)
    ap.add_argument("--max_iters", type=int, default=0)
    ap.add_argument("--save_epoch_freq", type=int, default=10)
    args = ap.parse_args(argv)
    rank, local, world = init_distributed()
    dev = "cuda:%d" % local
    opt = networks.default_options(ngf=args.ngf, ndf=args.ndf, lr=args.lr, no_TTUR=args.no_TTUR)
    tr = Trainer(opt, device=dev, world=world)
    save_dir = os.path.join(args.checkpoints_dir, args.name)
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    it = 0
    for epoch in range(1, args.niter + args.niter_decay + 1):
        for i in range(args.iters_per_epoch):
            batch = data.projector_batch(args.batchSize, dev, seed=1234 + rank + 977 * it)
            tr.step(batch)
            it += 1
            if rank == 0 and it % 10 == 0:
                print("(epoch: %d, iters: %d) " % (epoch, it)
                      + " ".join("%s: %.3f" % (k, float(v.mean())) for k, v in tr.get_latest_losses().items()))
            if args.max_iters and it >= args.max_iters:
                break
        tr.update_learning_rate(epoch, args.niter, args.niter_decay)
        if rank == 0 and (epoch % args.save_epoch_freq == 0 or (args.max_iters and it >= args.max_iters)):
            torch.save(tr.model.netG.state_dict(), os.path.join(save_dir, "latest_net_G.pth"))
            torch.save(tr.model.netD.state_dict(), os.path.join(save_dir, "latest_net_D.pth"))
        if args.max_iters and it >= args.max_iters:
            break


if __name__ == "__main__":
    main()
