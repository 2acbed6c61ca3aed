"""Entry point mirroring ``GenProjector/train.py``: per iteration one generator step and one discriminator step
(``train.py:28-37``), torchrun-aware.  ``--synthetic`` batches replace the licence-restricted Laval dataset.

    python -m emlight_amd.GenProjector.train --synthetic --batchSize 8 --max_iters 10
    python -m emlight_amd.GenProjector.train --synthetic --name lavalindoor --dataset_mode lavalindoor --dataroot /data/laval \
        --display_freq 1000 --batchSize 16 --niter 100 --niter_decay 100 --gpu_ids 0 --continue_train      # train_laval.sh's flags
    python -m emlight_amd.GenProjector.train --synthetic --continue_train --which_epoch latest     # resume (iter.txt)
    torchrun --nproc-per-node 8 -m emlight_amd.GenProjector.train --synthetic
"""
import argparse
import os

import torch

from ..RegressionNetwork.engine import init_distributed
from . import data, networks, options
from .iter_counter import IterationCounter
from .model_trainer import Trainer


def parse_args(argv=None):
    """The reference's flags (``options/train_options.py`` over ``base_options.py``: ``train_laval.sh`` runs unchanged --
    see ``options.py`` for what each one means here) plus ``--synthetic --iters_per_epoch --max_iters`` and the VGG switches.
    Needs no GPU: a launch that disagrees with ``--gpu_ids``, or a dataset the package cannot read, exits here."""
    ap = options.train_parser()
    networks.add_vgg_arguments(ap)
    args = ap.parse_args(argv)
    options.resolve_gpu_ids(args.gpu_ids, options.world_from_env())
    args.ignored_reference_flags = options.check_data_flags(args, ap, args.synthetic,
                                                            verbose=int(os.environ.get("RANK", "0")) == 0)
    return args


def main(argv=None):
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()   # kernel arguments in device memory, recorded library-GEMM selection: an entry point's choice
    args = parse_args(argv)       # before any process group exists: a wrong launch exits at once
    rank, local, world = init_distributed()
    dev = "cuda:%d" % local
    opt = options.network_options(args, True, **networks.vgg_options(args, verbose=rank == 0))
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
            if i % args.D_steps_per_G == 0:          # train.py:33-37
                tr.run_generator_one_step(batch)
            tr.run_discriminator_one_step(batch)
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
