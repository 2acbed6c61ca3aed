"""The projector's entry points accept the reference's own command lines (missing #3):
``/root/reference/GenProjector/train_laval.sh:1-10`` and ``test.sh:1-6`` -- the argv below is those scripts' verbatim --
through parsers that carry every flag of ``options/base_options.py:20-70``, ``train_options.py:11-46``, ``test_options.py:11-22``
with the reference's defaults.  CPU only: nothing here touches a GPU."""
import sys

import pytest

from emlight_amd.GenProjector import options
from emlight_amd.GenProjector import test as gp_test
from emlight_amd.GenProjector import train as gp_train

TRAIN_LAVAL_SH = ["--name", "lavalindoor", "--dataset_mode", "lavalindoor", "--dataroot",
                  "/home/fangneng.zfn/datasets/LavalIndoor/tpami/", "--display_freq", "1000", "--batchSize", "16", "--niter", "100",
                  "--niter_decay", "100", "--gpu_ids", "0,1", "--continue_train"]
TEST_SH = ["--name", "lavalindoor", "--checkpoints_dir", "./checkpoints", "--which_epoch", "100", "--dataset_mode", "lavalindoor",
           "--dataroot", "/home/fangneng.zfn/datasets/LavalIndoor/test/"]

# (flag, reference default) of base_options.py:20-70 / train_options.py:11-46 / generator.py:19-25 / discriminator.py:18-28,71-74
REFERENCE_TRAIN_DEFAULTS = dict(
    name="label2coco", gpu_ids="0", checkpoints_dir="./checkpoints", model="pix2pix", norm_G="spectralspadesyncbatch3x3",
    norm_D="spectralinstance", norm_E="spectralinstance", phase="train", batchSize=4, preprocess_mode="resize_and_crop",
    load_size=256, crop_size=256, aspect_ratio=2.0, label_nc=3, contain_dontcare_label=False, output_nc=3,
    dataroot="/home/fangneng.zfn/datasets/LavalIndoor/nips/", dataset_mode="coco", serial_batches=False, no_flip=False,
    nThreads=0, max_dataset_size=sys.maxsize, load_from_opt_file=False, cache_filelist_write=False, cache_filelist_read=False,
    display_winsize=400, netG="spade", ngf=64, init_type="xavier", init_variance=0.02, z_dim=256, no_instance=False, nef=16,
    use_vae=False, num_upsampling_layers="normal", display_freq=1000, print_freq=1000, save_latest_freq=1000,
    save_epoch_freq=10, no_html=False, debug=False, tf_log=False, continue_train=False, which_epoch="latest", niter=250,
    niter_decay=0, optimizer="adam", no_TTUR=False, lr=0.0002, D_steps_per_G=1, ndf=64, lambda_feat=10.0, lambda_vgg=10.0,
    no_ganFeat_loss=False, no_vgg_loss=False, gan_mode="hinge", netD="multiscale", lambda_kld=0.05, netD_subarch="n_layer",
    num_D=2, n_layers_D=4)


def test_every_reference_train_flag_exists_with_the_reference_default():
    ap = options.train_parser()
    from emlight_amd.GenProjector import networks
    networks.add_vgg_arguments(ap)
    for k, v in REFERENCE_TRAIN_DEFAULTS.items():
        assert ap.get_default(k) == v, (k, ap.get_default(k), v)
    # train_options.py:31-35: the betas' defaults follow --no_TTUR
    o = options.network_options(ap.parse_args([]), True)
    assert (o.beta1, o.beta2) == (0.0, 0.9)
    o = options.network_options(ap.parse_args(["--no_TTUR"]), True)
    assert (o.beta1, o.beta2) == (0.5, 0.999)
    o = options.network_options(ap.parse_args(["--no_TTUR", "--beta1", "0.3"]), True)
    assert (o.beta1, o.beta2) == (0.3, 0.999)


def test_train_laval_sh_argv_under_a_two_rank_launch(monkeypatch, capsys):
    """``--gpu_ids 0,1`` = two GPUs = two ranks here; the dataset flags are accepted and named as ignored with --synthetic."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    args = gp_train.parse_args(TRAIN_LAVAL_SH + ["--synthetic", "--max_iters", "0"])
    assert (args.name, args.batchSize, args.niter, args.niter_decay, args.continue_train) == ("lavalindoor", 16, 100, 100, True)
    assert args.dataset_mode == "lavalindoor" and args.display_freq == 1000
    assert set(args.ignored_reference_flags) == {"dataset_mode", "dataroot"}      # display_freq 1000 IS the default
    assert "accepted and ignored" in capsys.readouterr().out
    opt = options.network_options(args, True, no_vgg_loss=True)
    assert (opt.ngf, opt.ndf, opt.num_D, opt.n_layers_D, opt.lr) == (64, 64, 2, 4, 0.0002)


def test_train_laval_sh_argv_in_one_process_is_refused_with_the_launch_command(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        gp_train.parse_args(TRAIN_LAVAL_SH + ["--synthetic"])
    assert "torchrun" in str(e.value) and "--nproc-per-node 2" in str(e.value)
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        gp_train.parse_args(TRAIN_LAVAL_SH + ["--synthetic"])
    assert "must agree" in str(e.value)
    # the reference's default "--gpu_ids 0" under torchrun: fine at any world size
    assert gp_train.parse_args(["--synthetic"]).gpu_ids == "0"
    with pytest.raises(SystemExit) as e:
        gp_train.parse_args(["--synthetic", "--gpu_ids", "-1"])
    assert "no CPU path" in str(e.value)


def test_the_dataset_reader_is_refused_by_name_without_synthetic(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        gp_train.parse_args(TRAIN_LAVAL_SH)
    assert "--synthetic" in str(e.value) and "lavalindoor" in str(e.value)


def test_test_sh_argv(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = gp_test.parse_args(TEST_SH + ["--synthetic"])
    assert (args.name, args.which_epoch, args.checkpoints_dir, args.results_dir) == ("lavalindoor", "100", "./checkpoints", "./results/")
    # test_options.py:16-20
    assert (args.preprocess_mode, args.serial_batches, args.no_flip, args.phase) == ("scale_width_and_crop", True, True, "test")
    assert args.gpu_id_list == [0] and args.how_many == float("inf")
    opt = options.network_options(args, False)
    assert opt.isTrain is False and opt.ngf == 64


def test_unknown_architectures_are_refused():
    ap = options.train_parser()
    with pytest.raises(SystemExit):
        options.network_options(ap.parse_args(["--netG", "pix2pixhd"]), True)
