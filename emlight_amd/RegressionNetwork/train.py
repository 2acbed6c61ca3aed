"""Entry point mirroring ``RegressionNetwork/train.py`` (the reference file has unresolved
merge-conflict markers and module-level constants; this one is argparse-driven and
torchrun-aware, same optimiser, loss weights, print/visualise/checkpoint cadence).

    python -m emlight_amd.RegressionNetwork.train --synthetic --epochs 1
    torchrun --nproc-per-node 8 -m emlight_amd.RegressionNetwork.train --synthetic
"""
import argparse
import os

import numpy as np
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import data, util
from .engine import RegressionTrainer, init_distributed


def save_visual(path, crop, pred, gt, ln, tone):
    """``train.py:110-145``: GT lobes over predicted lobes next to the crop (needs PIL)."""
    from PIL import Image
    dev = crop.device
    dirs = torch.from_numpy(util.sphere_points(ln)).float().view(1, ln * 3).to(dev)
    size = torch.full((1, ln), 0.0025, device=dev)
    rows = []
    for p in (gt, pred):
        inten = p["intensity"][0].view(1, 1, 1).repeat(1, ln, 3) * 500
        dist_ = p["distribution"][0].view(1, ln, 1).repeat(1, 1, 3)
        rgb = p["rgb_ratio"][0].view(1, 1, 3).repeat(1, ln, 1)
        env = util.convert_to_panorama(dirs, size, (dist_ * inten * rgb).reshape(1, ln * 3).contiguous())
        env = np.squeeze(env[0].detach().cpu().numpy())
        rows.append(tone(env)[0].transpose((1, 2, 0)).astype("float32") * 255.0)
    env = Image.fromarray(np.vstack(rows).astype("uint8")).resize((256, 256))
    c = Image.fromarray((crop[0].detach().cpu().numpy().transpose((1, 2, 0)) * 255.0).astype("uint8")).resize((256, 256))
    Image.fromarray(np.hstack((np.array(c), np.array(env))).astype("uint8")).save(path)


def main(argv=None):
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()   # kernel arguments in device memory, recorded library-GEMM selection: an entry point's choice
    ap = argparse.ArgumentParser()
    ap.add_argument("--train_dir", default=None, help="directory in PickleParameterDataset format")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--batch_size", type=int, default=16, help="per-GPU batch (reference: 16)")
    ap.add_argument("--anchors", type=int, default=96)
    ap.add_argument("--crop_hw", type=int, nargs=2, default=(192, 256))
    ap.add_argument("--blur", type=float, default=.025)
    ap.add_argument("--diameter", type=float, default=None)
    ap.add_argument("--sync_diameter", type=int, choices=(0, 1), default=None,
                    help="Sinkhorn eps-schedule from the range of the GLOBAL batch (2-float all-reduce per step); default: "
                         "on under torchrun unless --diameter is given (a single-process run sees the whole batch)")
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--max_iters", type=int, default=0)
    ap.add_argument("--save_dir", default="./checkpoints")
    ap.add_argument("--summary_dir", default="./summary")
    ap.add_argument("--load", default=None, help="state_dict to resume from (reference .pth files load)")
    args = ap.parse_args(argv)

    rank, local, world = init_distributed()
    device = "cuda:%d" % local
    tr = RegressionTrainer(anchors=args.anchors, crop_hw=tuple(args.crop_hw), blur=args.blur,
                           diameter=args.diameter, device=device, world=world,
                           sync_diameter=None if args.sync_diameter is None else bool(args.sync_diameter))
    if args.load:
        tr.model.load_state_dict(torch.load(args.load, map_location=device))
        if rank == 0:
            print("load trained model")
    if args.synthetic or not args.train_dir:
        ds = data.SyntheticParameterDataset(anchors=args.anchors, crop_hw=tuple(args.crop_hw))
    else:
        ds = data.PickleParameterDataset(args.train_dir)
    sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True) if world > 1 else None
    loader = DataLoader(ds, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                        drop_last=True, num_workers=2, pin_memory=True)
    tone = util.TonemapHDR(gamma=2.4, percentile=99, max_mapping=0.99)
    if rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        os.makedirs(args.summary_dir, exist_ok=True)
    it = 0
    for epoch in range(args.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        if rank == 0:
            print("{} optim: {}".format(epoch, tr.optimizer.param_groups[0]["lr"]))
        for i, para in enumerate(loader):
            batch = {k: v.to(device, non_blocking=True) for k, v in para.items() if k != "name"}
            loss, terms = tr.step(batch)
            if rank == 0 and i % 10 == 0:  # the only host syncs (train.py:106-108)
                print("epoch {:0>3d} batch {:0>3d}, ".format(epoch, i)
                      + ", ".join("{}:{}".format(k, v.item()) for k, v in terms.items()))
            if rank == 0 and i % 100 == 0:
                # the reference renders the prediction of THIS training step (train.py:110-133), no extra forward
                pred = {k: v.detach() for k, v in tr.last_pred.items()}
                try:
                    save_visual(os.path.join(args.summary_dir, "{}_{}.jpg".format(epoch, i)),
                                batch["crop"], pred, batch, args.anchors, tone)
                except ImportError:
                    pass
            if rank == 0 and i % 500 == 0:
                print("saving the latest model")
                torch.save(tr.model.state_dict(), os.path.join(args.save_dir, "latest_net.pth"))
            it += 1
            if args.max_iters and it >= args.max_iters:
                break
        if rank == 0 and epoch % 10 == 0:
            print("saving the model at the end of epoch %d" % epoch)
            torch.save(tr.model.state_dict(), os.path.join(args.save_dir, "%s_net.pth" % epoch))
            torch.save(tr.model.state_dict(), os.path.join(args.save_dir, "latest_net.pth"))
        if args.max_iters and it >= args.max_iters:
            break


if __name__ == "__main__":
    main()
