"""Entry point mirroring ``GenProjector/train.py``: per iteration one generator step and one discriminator step
(``train.py:28-37``), torchrun-aware.  ``--synthetic`` batches replace the licence-restricted Laval dataset.

    python -m emlight_amd.GenProjector.train --synthetic --batchSize 8 --max_iters 10
    python -m emlight_amd.GenProjector.train --synthetic --continue_train --which_epoch latest     # resume (iter.txt)
    torchrun --nproc-per-node 8 -m emlight_amd.GenProjector.train --synthetic
"""
import argparse
import os

import torch

from ..RegressionNetwork.engine import init_distributed
from . import data, networks
from .iter_counter import IterationCounter
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
    ap.add_argument("--save_latest_freq", type=int, default=5000, help="in samples, like the reference")
    ap.add_argument("--print_freq", type=int, default=100, help="in samples, like the reference")
    ap.add_argument("--continue_train", action="store_true", help="resume from <which_epoch>_net_{G,D}.pth and iter.txt")
    ap.add_argument("--which_epoch", default="latest")
    networks.add_vgg_arguments(ap)
    args = ap.parse_args(argv)
    rank, local, world = init_distributed()
    dev = "cuda:%d" % local
    opt = networks.default_options(ngf=args.ngf, ndf=args.ndf, lr=args.lr, no_TTUR=args.no_TTUR,
                                   **networks.vgg_options(args, verbose=rank == 0))
    tr = Trainer(opt, device=dev, world=world)
    save_dir = os.path.join(args.checkpoints_dir, args.name)
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "objective.txt"), "w") as fh:   # which objective these checkpoints were trained on
            fh.write("vgg_term: %s\n" % tr.model.vgg_variant)
    if args.continue_train:
        tr.load(args.which_epoch, save_dir)      # every rank loads the same files: replicas start identical
    global_batch = args.batchSize * world
    counter = IterationCounter(args.checkpoints_dir, args.name, args.iters_per_epoch * global_batch, global_batch, args.niter,
                               args.niter_decay, args.continue_train, args.print_freq, args.save_latest_freq,
                               args.save_epoch_freq)
    it = 0
    for epoch in counter.training_epochs():
        counter.record_epoch_start(epoch)
        for i in range(counter.epoch_iter // global_batch, args.iters_per_epoch):
            counter.record_one_iteration()
            batch = data.projector_batch(args.batchSize, dev, seed=1234 + rank + 977 * (counter.total_steps_so_far // global_batch))
            tr.step(batch)
            it += 1
            if rank == 0 and counter.needs_printing():   # the only host syncs
                print("(epoch: %d, iters: %d, time: %.3f) " % (epoch, counter.epoch_iter, counter.time_per_iter)
                      + " ".join("%s: %.3f" % (k, float(v.mean())) for k, v in tr.get_latest_losses().items()))
            if rank == 0 and counter.needs_saving():
                print("saving the latest model (epoch %d, total_steps %d)" % (epoch, counter.total_steps_so_far))
                tr.save("latest", save_dir)
                counter.record_current_iter()
            if args.max_iters and it >= args.max_iters:
                break
        tr.update_learning_rate(epoch, args.niter, args.niter_decay)
        counter.record_epoch_end(write=rank == 0)
        stop = bool(args.max_iters and it >= args.max_iters)
        if rank == 0 and (epoch % args.save_epoch_freq == 0 or epoch == counter.total_epochs or stop):
            print("saving the model at the end of epoch %d, iters %d" % (epoch, counter.total_steps_so_far))
            tr.save("latest", save_dir)
            tr.save(epoch, save_dir)
            if stop:
                counter.record_current_iter()
        if stop:
            break


if __name__ == "__main__":
    main()
