"""Entry point mirroring the intent of ``RegressionNetwork/test.py`` (the reference file is
broken: undefined ``train_dir``, 42 anchors against ln=96): load ``latest_net.pth``, run the
encoder, dump ``{distribution, rgb_ratio, intensity*500}`` per image to
``./results/<name>.pickle`` (``test.py:79-85``)."""
import argparse
import os
import pickle

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import data
from .DenseNet import DenseNet


def main(argv=None):
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()   # kernel arguments in device memory, recorded library-GEMM selection: an entry point's choice
    ap = argparse.ArgumentParser()
    ap.add_argument("--test_dir", default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--checkpoint", default="./checkpoints/latest_net.pth")
    ap.add_argument("--anchors", type=int, default=96)
    ap.add_argument("--crop_hw", type=int, nargs=2, default=(192, 256))
    ap.add_argument("--results_dir", default="./results")
    ap.add_argument("--max_images", type=int, default=100)
    args = ap.parse_args(argv)

    device = torch.device("cuda")
    model = DenseNet(anchors=args.anchors, crop_hw=tuple(args.crop_hw)).to(device)
    if args.checkpoint and os.path.exists(args.checkpoint):
        model.load_state_dict(torch.load(args.checkpoint, map_location=device))
        print("load trained model")
    # the reference never calls .eval() (test.py:36-37): BN uses batch statistics at B=1.
    if args.synthetic or not args.test_dir:
        ds = data.SyntheticParameterDataset(length=args.max_images, anchors=args.anchors, crop_hw=tuple(args.crop_hw))
    else:
        ds = data.PickleParameterDataset(args.test_dir)
    os.makedirs(args.results_dir, exist_ok=True)
    ln = args.anchors
    with torch.no_grad():
        for i, para in enumerate(DataLoader(ds, batch_size=1, shuffle=False)):
            if i >= args.max_images:
                break
            pred = model(para["crop"].to(device))
            out = {"distribution": np.squeeze(pred["distribution"][0].view(ln).cpu().numpy()),
                   "rgb_ratio": np.squeeze(pred["rgb_ratio"][0].view(3).cpu().numpy()),
                   "intensity": np.squeeze((pred["intensity"][0] * 500).cpu().numpy())}
            with open(os.path.join(args.results_dir, para["name"][0] + ".pickle"), "wb") as f:
                pickle.dump(out, f, protocol=pickle.HIGHEST_PROTOCOL)
            print(i)


if __name__ == "__main__":
    main()
