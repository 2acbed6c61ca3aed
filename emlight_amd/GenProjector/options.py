"""The projector's command line (reference ``GenProjector/options/base_options.py:20-70``, ``train_options.py:11-46``,
``test_options.py:11-22`` and the networks' ``modify_commandline_options``: ``generator.py:19-25``,
``discriminator.py:18-28,71-74``): every flag of the reference's parsers is ACCEPTED with the reference's name, type and
default, so that ``train_laval.sh`` / ``test.sh`` run against this package unchanged.  What a flag means here:

* network / optimiser / schedule / checkpoint flags: honoured (they build the same ``opt`` namespace the modules read);
* ``--gpu_ids``: the reference runs ONE process that drives the listed GPUs through DataParallel; here every GPU is its own
  process under torchrun.  ``resolve_gpu_ids`` turns the list into the launch it stands for: it must agree with
  ``WORLD_SIZE`` (one process: one id; N ranks: N ids or the reference default "0"), and a multi-id list in a single process
  is refused with the torchrun command that runs it -- never silently trained on one GPU;
* dataset / display flags (``--dataset_mode --dataroot --preprocess_mode --display_freq --no_html ...``): the Laval dataset
  reader and the HTML visualiser are out of scope (the data loader is the caller's); with ``--synthetic`` they
  are accepted and ignored (one line says which), without it a non-default ``--dataset_mode`` is an error that says so.

Options of this build that the reference does not have: ``--synthetic --iters_per_epoch --max_iters`` and the VGG switches.
"""
import argparse
import os
import sys

# flags that select or configure the reference's dataset reader / visualiser: accepted, never used by the kernels' path
IGNORED_DATA_FLAGS = ("dataset_mode", "dataroot", "preprocess_mode", "load_size", "serial_batches", "no_flip", "nThreads",
                      "max_dataset_size", "load_from_opt_file", "cache_filelist_write", "cache_filelist_read",
                      "display_winsize", "display_freq", "no_html", "debug", "tf_log", "phase", "no_instance", "nef", "use_vae",
                      "z_dim", "contain_dontcare_label")


def add_base_options(ap):
    """``BaseOptions.initialize`` (base_options.py:20-70): same names, types, defaults."""
    ap.add_argument("--name", type=str, default="label2coco", help="name of the experiment: where checkpoints are stored")
    ap.add_argument("--gpu_ids", type=str, default="0", help="reference: GPUs of the one DataParallel process; here it must "
                    "agree with the torchrun launch (see module docstring)")
    ap.add_argument("--checkpoints_dir", type=str, default="./checkpoints")
    ap.add_argument("--model", type=str, default="pix2pix")
    ap.add_argument("--norm_G", type=str, default="spectralspadesyncbatch3x3")     # generator.py:21 set_defaults
    ap.add_argument("--norm_D", type=str, default="spectralinstance")
    ap.add_argument("--norm_E", type=str, default="spectralinstance")
    ap.add_argument("--phase", type=str, default="train")
    ap.add_argument("--batchSize", type=int, default=4, help="reference: over all --gpu_ids; here: per process (per GPU)")
    ap.add_argument("--preprocess_mode", type=str, default="resize_and_crop")
    ap.add_argument("--load_size", type=int, default=256)
    ap.add_argument("--crop_size", type=int, default=256)
    ap.add_argument("--aspect_ratio", type=float, default=2.0)
    ap.add_argument("--label_nc", type=int, default=3)
    ap.add_argument("--contain_dontcare_label", action="store_true")
    ap.add_argument("--output_nc", type=int, default=3)
    ap.add_argument("--dataroot", type=str, default="/home/fangneng.zfn/datasets/LavalIndoor/nips/")
    ap.add_argument("--dataset_mode", type=str, default="coco")
    ap.add_argument("--serial_batches", action="store_true")
    ap.add_argument("--no_flip", action="store_true")
    ap.add_argument("--nThreads", default=0, type=int)
    ap.add_argument("--max_dataset_size", type=int, default=sys.maxsize)
    ap.add_argument("--load_from_opt_file", action="store_true")
    ap.add_argument("--cache_filelist_write", action="store_true")
    ap.add_argument("--cache_filelist_read", action="store_true")
    ap.add_argument("--display_winsize", type=int, default=400)
    ap.add_argument("--netG", type=str, default="spade")
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--init_type", type=str, default="xavier")
    ap.add_argument("--init_variance", type=float, default=0.02)
    ap.add_argument("--z_dim", type=int, default=256)
    ap.add_argument("--no_instance", action="store_true")
    ap.add_argument("--nef", type=int, default=16)
    ap.add_argument("--use_vae", action="store_true")
    # generator.py:19-25
    ap.add_argument("--num_upsampling_layers", choices=("normal", "more", "most"), default="normal")
    return ap


def add_train_options(ap):
    """``TrainOptions.initialize`` (train_options.py:11-46) + the discriminators' options (discriminator.py:18-28,71-74)."""
    add_base_options(ap)
    ap.add_argument("--display_freq", type=int, default=1000)
    ap.add_argument("--print_freq", type=int, default=1000, help="in samples, like the reference")
    ap.add_argument("--save_latest_freq", type=int, default=1000, help="in samples, like the reference")
    ap.add_argument("--save_epoch_freq", type=int, default=10)
    ap.add_argument("--no_html", action="store_true")
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--tf_log", action="store_true")
    ap.add_argument("--continue_train", action="store_true", help="resume from <which_epoch>_net_{G,D}.pth and iter.txt")
    ap.add_argument("--which_epoch", type=str, default="latest")
    ap.add_argument("--niter", type=int, default=250)
    ap.add_argument("--niter_decay", type=int, default=0)
    ap.add_argument("--optimizer", type=str, default="adam")
    ap.add_argument("--beta1", type=float, default=None, help="default 0.0 (0.5 with --no_TTUR), as train_options.py:31-35")
    ap.add_argument("--beta2", type=float, default=None, help="default 0.9 (0.999 with --no_TTUR)")
    ap.add_argument("--no_TTUR", action="store_true")
    ap.add_argument("--lr", type=float, default=0.0002)
    ap.add_argument("--D_steps_per_G", type=int, default=1)
    ap.add_argument("--ndf", type=int, default=64)
    ap.add_argument("--lambda_feat", type=float, default=10.0)
    ap.add_argument("--lambda_vgg", type=float, default=10.0)
    ap.add_argument("--no_ganFeat_loss", action="store_true")
    ap.add_argument("--gan_mode", type=str, default="hinge")
    ap.add_argument("--netD", type=str, default="multiscale")
    ap.add_argument("--lambda_kld", type=float, default=0.05)
    ap.add_argument("--netD_subarch", type=str, default="n_layer")
    ap.add_argument("--num_D", type=int, default=2)
    ap.add_argument("--n_layers_D", type=int, default=4)
    return ap


def add_test_options(ap):
    """``TestOptions.initialize`` (test_options.py:11-22)."""
    add_base_options(ap)
    ap.add_argument("--results_dir", type=str, default="./results/")
    ap.add_argument("--which_epoch", type=str, default="latest")
    ap.add_argument("--how_many", type=float, default=float("inf"), help="how many test batches to run")
    ap.set_defaults(preprocess_mode="scale_width_and_crop", crop_size=256, load_size=256, display_winsize=256,
                    serial_batches=True, no_flip=True, phase="test")
    return ap


def resolve_gpu_ids(gpu_ids, world, local_rank=0):
    """``--gpu_ids`` (base_options.py:161-170: the GPUs of the reference's one process) against the process-per-GPU launch.
    Returns the list of ids; raises ``SystemExit`` with the command to run when the two disagree."""
    try:
        ids = [int(s) for s in str(gpu_ids).split(",") if s.strip() != ""]
    except ValueError:
        raise SystemExit("--gpu_ids %r: want a comma-separated list of integers (e.g. 0 or 0,1)" % (gpu_ids,))
    ids = [i for i in ids if i >= 0]
    if not ids:
        raise SystemExit("--gpu_ids -1 (CPU) is not available: the projector runs on the MI355X kernels only, there is no "
                         "CPU path")
    if len(ids) == world or (world > 1 and ids == [0]):   # "0" is the reference's default: a torchrun launch may leave it
        return ids
    if world == 1:
        raise SystemExit("--gpu_ids %s names %d GPUs but this is a single process: the reference's DataParallel is one "
                         "process per GPU here -- launch it as\n    torchrun --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 -m emlight_amd.GenProjector.train <the same arguments>\n(--batchSize is "
                         "then per GPU; the reference's is over all of them)" % (gpu_ids, len(ids), len(ids)))
    raise SystemExit("--gpu_ids %s names %d GPU(s) but the launcher started %d rank(s) (WORLD_SIZE): they must agree"
                     % (gpu_ids, len(ids), world))


def check_data_flags(args, ap, synthetic, verbose=True):
    """Dataset / display flags: ignored (and named, once) on the synthetic path, refused otherwise."""
    given = [k for k in IGNORED_DATA_FLAGS if hasattr(args, k) and getattr(args, k) != ap.get_default(k)]
    if synthetic:
        if given and verbose:
            print("GenProjector: --synthetic batches -- these reference dataset / display options are accepted and "
                  "ignored: %s" % ", ".join("--" + k for k in given))
        return given
    raise SystemExit("GenProjector: the Laval dataset reader (--dataset_mode %s, --dataroot %s) is outside this package "
                     "(SURVEY 8b: the data loader is the caller's): pass --synthetic for the seeded synthetic batches of "
                     "SURVEY 8d, or feed `Trainer.step` your own batches {input, crop, warped, map}"
                     % (getattr(args, "dataset_mode", "?"), getattr(args, "dataroot", "?")))


def network_options(args, is_train=True, **extra):
    """The ``opt`` namespace the networks read (``networks.default_options``) from the parsed reference flags."""
    from . import networks
    kw = dict(ngf=args.ngf, crop_size=args.crop_size, aspect_ratio=args.aspect_ratio,
              num_upsampling_layers=args.num_upsampling_layers, norm_G=args.norm_G, norm_D=args.norm_D, norm_E=args.norm_E,
              label_nc=args.label_nc, output_nc=args.output_nc, semantic_nc=args.label_nc, init_type=args.init_type,
              init_variance=args.init_variance, isTrain=bool(is_train))
    if is_train:
        # train_options.py:31-35: the Adam betas' defaults depend on --no_TTUR
        b1 = args.beta1 if args.beta1 is not None else (0.5 if args.no_TTUR else 0.0)
        b2 = args.beta2 if args.beta2 is not None else (0.999 if args.no_TTUR else 0.9)
        kw.update(ndf=args.ndf, num_D=args.num_D, n_layers_D=args.n_layers_D, netD_subarch=args.netD_subarch,
                  no_ganFeat_loss=args.no_ganFeat_loss, gan_mode=args.gan_mode, lr=args.lr, beta1=b1, beta2=b2,
                  no_TTUR=args.no_TTUR)
    kw.update(extra)
    for name, want in (("model", "pix2pix"), ("netG", "spade")) + ((("netD", "multiscale"), ("netD_subarch", "n_layer"),
                                                                    ("optimizer", "adam")) if is_train else ()):
        if getattr(args, name) != want:
            raise SystemExit("--%s %s: only %r exists in the reference's projector (and here)" % (name, getattr(args, name), want))
    return networks.default_options(**kw)


def train_parser():
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    add_train_options(ap)
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic batches (SURVEY 8d) instead of the Laval dataset")
    ap.add_argument("--iters_per_epoch", type=int, default=100, help="synthetic: iterations that make an epoch")
    ap.add_argument("--max_iters", type=int, default=0, help="stop after this many iterations (0: run niter + niter_decay epochs)")
    return ap


def test_parser():
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    add_test_options(ap)
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic batches (SURVEY 8d) instead of the Laval dataset")
    return ap


def world_from_env():
    return int(os.environ.get("WORLD_SIZE", "1"))
