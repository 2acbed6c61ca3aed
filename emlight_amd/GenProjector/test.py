"""Entry point mirroring ``GenProjector/test.py:20-38``: load ``<which_epoch>_net_G.pth``, run the generator in
inference mode on each batch, write the predicted HDR panoramas as ``.npy`` (EXR writing is out of scope)."""
import argparse
import os

import numpy as np
import torch

from . import data, networks
from .pix2pix_model import Pix2PixModel


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="laval")
    ap.add_argument("--checkpoints_dir", default="./checkpoints")
    ap.add_argument("--which_epoch", default="latest")
    ap.add_argument("--results_dir", default="./results")
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--batchSize", type=int, default=1)
    ap.add_argument("--how_many", type=int, default=10)
    args = ap.parse_args(argv)
    dev = "cuda"
    opt = networks.default_options(ngf=args.ngf, isTrain=False)
    model = Pix2PixModel(opt).to(dev).eval()
    path = os.path.join(args.checkpoints_dir, args.name, "%s_net_G.pth" % args.which_epoch)
    if os.path.exists(path):
        model.netG.load_state_dict(torch.load(path, map_location=dev))
    os.makedirs(args.results_dir, exist_ok=True)
    for i in range(args.how_many):
        batch = data.projector_batch(args.batchSize, dev, seed=4321 + i)
        fake = model(batch, mode="inference")
        np.save(os.path.join(args.results_dir, "pred_%04d.npy" % i), fake.cpu().numpy())
        print("process image... %d" % i)


if __name__ == "__main__":
    main()
