"""Entry point mirroring ``GenProjector/test.py:20-38``: load ``<which_epoch>_net_G.pth``, run the generator in
inference mode on each batch, write the predicted HDR panoramas as ``.npy`` (EXR writing is out of scope)."""
import argparse
import os

import numpy as np
import torch

from . import data, networks, options
from .pix2pix_model import Pix2PixModel


def parse_args(argv=None):
    """The reference's flags (``options/test_options.py`` over ``base_options.py``: ``test.sh`` runs unchanged) + ``--synthetic``."""
    ap = options.test_parser()
    args = ap.parse_args(argv)
    args.gpu_id_list = options.resolve_gpu_ids(args.gpu_ids, 1)
    args.ignored_reference_flags = options.check_data_flags(args, ap, args.synthetic)
    return args


def main(argv=None):
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()   # kernel arguments in device memory, recorded library-GEMM selection: an entry point's choice
    args = parse_args(argv)
    dev = "cuda:%d" % args.gpu_id_list[0]
    opt = options.network_options(args, False)
    model = Pix2PixModel(opt).to(dev).eval()
    path = os.path.join(args.checkpoints_dir, args.name, "%s_net_G.pth" % args.which_epoch)
    if os.path.exists(path):
        model.netG.load_state_dict(torch.load(path, map_location=dev))
    os.makedirs(args.results_dir, exist_ok=True)
    # the reference stops after 1000 samples (test.py:23-25); the synthetic stream is endless, so --how_many bounds it (default 10)
    how_many = 10 if args.how_many == float("inf") else int(args.how_many)
    for i in range(how_many):
        batch = data.projector_batch(args.batchSize, dev, seed=4321 + i)
        fake = model(batch, mode="inference")
        np.save(os.path.join(args.results_dir, "pred_%04d.npy" % i), fake.cpu().numpy())
        print("process image... %d" % i)


if __name__ == "__main__":
    main()
