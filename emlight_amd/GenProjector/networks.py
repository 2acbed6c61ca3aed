"""SPADE generator, conv encoder, multiscale PatchGAN discriminator and GAN losses of the projector.

Clean restatements of ``GenProjector/models/networks/{generator,architecture,normalization,discriminator,
loss}.py`` with the same module names (hence the same ``state_dict`` keys: reference ``*_net_G.pth`` /
``*_net_D.pth`` load) and forward signatures.  ``opt`` is any namespace with the reference's option names;
``default_options()`` gives the reference defaults (``options/base_options.py``, ``train_options.py``).
"""
from argparse import Namespace

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import spherenet
from .._knobs import knob_flag
from .spherenet import SphereConv2D


def default_options(**kw):
    opt = Namespace(ngf=64, ndf=64, crop_size=256, aspect_ratio=2.0, num_upsampling_layers="normal",
                    norm_G="spectralspadesyncbatch3x3", norm_D="spectralinstance", norm_E="spectralinstance",
                    label_nc=3, output_nc=3, semantic_nc=3, num_D=2, n_layers_D=4, netD_subarch="n_layer",
                    no_ganFeat_loss=False, no_vgg_loss=True, gan_mode="hinge", lr=2e-4, beta1=0.0, beta2=0.9,
                    no_TTUR=False, init_type="xavier", init_variance=0.02, isTrain=True)
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


def add_vgg_arguments(ap):
    """The perceptual-term switches of the training entry points (``GenProjector/train.py``, ``joint.py``)."""
    ap.add_argument("--vgg_weights", default=None,
                    help="torchvision vgg19 state dict (.pth): the reference's perceptual term (pix2pix_model.py:119-120)")
    ap.add_argument("--vgg_random", action="store_true",
                    help="run the term on seeded RANDOM features: the reference's work (timing), not its objective")
    ap.add_argument("--no_vgg_loss", action="store_true", help="drop the term (also the default when no weights are given)")


def vgg_options(args, verbose=True):
    """``default_options`` keywords from those switches.  The reference always adds the VGG term; its ImageNet weights
    cannot be obtained offline, so: weights given -> the reference's objective; ``--vgg_random`` -> explicit opt-in to
    random features; neither -> the term is OFF and the entry point says so (never a silent objective)."""
    on = not args.no_vgg_loss and (args.vgg_weights is not None or args.vgg_random)
    if verbose and not on and not args.no_vgg_loss:
        print("VGG perceptual term OFF: pass --vgg_weights <torchvision vgg19 .pth> for the reference's objective "
              "(or --vgg_random to time the term on random features)")
    return dict(no_vgg_loss=not on, vgg_weights=args.vgg_weights, vgg_random=bool(args.vgg_random))


def init_weights(net, init_type="xavier", gain=0.02):
    """``BaseNetwork.init_weights`` (``base_network.py:32-59``): xavier_normal(gain) on conv/linear weights."""
    def f(m):
        name = m.__class__.__name__
        if name.find("BatchNorm2d") != -1:
            if getattr(m, "weight", None) is not None:
                nn.init.normal_(m.weight.data, 1.0, gain)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif hasattr(m, "weight") and (name.find("Conv") != -1 or name.find("Linear") != -1):
            if init_type == "normal":
                nn.init.normal_(m.weight.data, 0.0, gain)
            elif init_type == "xavier":
                nn.init.xavier_normal_(m.weight.data, gain=gain)
            elif init_type == "kaiming":
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                nn.init.orthogonal_(m.weight.data, gain=gain)
            elif init_type != "none":
                raise NotImplementedError(init_type)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
    net.apply(f)
    return net


def nonspade_norm(norm_type):
    """``get_nonspade_norm_layer`` (``normalization.py:20-62``) for 'spectral' + {none, instance, batch}."""
    def wrap(layer):
        sub = norm_type
        if sub.startswith("spectral"):
            # the fused hook (csrc/spectral.hip: 5 launches instead of torch's ~15 per forward) serves every 3x3 convolution:
            # its result is the (O, C, 3, 3) weight in channels-last memory, which SphereConv2D's kernels AND MIOpen's NHWC
            # convolutions (the crop encoder's stride-2 nn.Conv2d layers) take as it is
            k3 = isinstance(layer, SphereConv2D) or (isinstance(layer, nn.Conv2d) and tuple(layer.kernel_size) == (3, 3)
                                                     and layer.groups == 1 and knob_flag("EML_SN_CONV2D", True))
            layer = spherenet.fused_spectral_norm(layer) if k3 else spectral_norm(layer)
            sub = sub[len("spectral"):]
        if sub in ("none", ""):
            return layer
        if getattr(layer, "bias", None) is not None:
            delattr(layer, "bias")
            layer.register_parameter("bias", None)
        ch = getattr(layer, "out_channels", None) or layer.weight.size(0)
        if sub == "instance":
            norm = nn.InstanceNorm2d(ch, affine=False)
        elif sub in ("batch", "sync_batch"):
            norm = nn.BatchNorm2d(ch, affine=True)
        else:
            raise ValueError("normalization layer %s is not recognized" % sub)
        return nn.Sequential(layer, norm)
    return wrap


def resized_guide(segmap, size):
    """``F.interpolate(segmap, size=size, mode="nearest")`` (normalization.py:106), computed once per guide map and size: the
    generator's 18 SPADE norms see 6 resolutions of the SAME map (30 launches fewer per pass, and in the joint step, where
    the map carries a gradient, 6 upsample backwards and gradient accumulations instead of 18).  The cache lives on the tensor
    object, so it dies with it; the full-resolution "resize" is the map itself (nearest to the same size copies)."""
    size = (int(size[0]), int(size[1]))
    if tuple(segmap.shape[2:]) == size:
        return segmap
    # a resize made under no_grad has no grad_fn: it must never serve a later pass that differentiates through the map
    #  -- the key carries whether this pass records a graph for the map
    key = (segmap._version, bool(torch.is_grad_enabled() and segmap.requires_grad))
    cache = getattr(segmap, "_eml_resized", None)
    if cache is None or cache[0] != key:
        cache = (key, {})
        segmap._eml_resized = cache
    if size not in cache[1]:
        cache[1][size] = F.interpolate(segmap, size=size, mode="nearest")
    return cache[1][size]


class SPADE(nn.Module):
    """``out = norm(x) * (1 + gamma(seg)) + beta(seg)`` (``normalization.py:68-115``).  The param-free norm is a
    BatchNorm2d(affine=False) module only as the holder of the running statistics (reference state_dict keys); its
    arithmetic is folded into the HIP modulation kernels, and with more than one rank its (2C+1)-float sums are
    all-reduced over RCCL -- the job of the reference's vendored ``sync_batchnorm`` package."""

    def __init__(self, config_text, norm_nc, label_nc):
        super().__init__()
        assert config_text.startswith("spade")
        kind = config_text[len("spade"):-3]
        if kind == "instance":
            self.param_free_norm = nn.InstanceNorm2d(norm_nc, affine=False)
        elif kind in ("syncbatch", "batch"):
            self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)
        else:
            raise ValueError("%s is not a recognized param-free norm type in SPADE" % kind)
        nhidden = 128
        self.mlp_shared = nn.Sequential(SphereConv2D(label_nc, nhidden), nn.ReLU())
        self.mlp_gamma = SphereConv2D(nhidden, norm_nc)
        self.mlp_beta = SphereConv2D(nhidden, norm_nc)

    def forward(self, x, segmap, slope=1.0, stats=None, up2=False):
        """``slope`` != 1 folds the LeakyReLU that follows this norm in SPADEResnetBlock (architecture.py:56-57) in;
        ``stats``: (mean, istd) of this x when a sibling norm already reduced it (norm_0 / norm_s share their input);
        ``up2``: ``x`` stands for its nearest x2 upsample (the generator's ``self.up``), which is never materialised."""
        size = (2 * x.size(2), 2 * x.size(3)) if up2 else x.size()[2:]
        segmap = resized_guide(segmap, size)
        conv, act = self.mlp_shared[0], self.mlp_shared[1]
        actv = conv(segmap, act_slope=0.0) if isinstance(act, nn.ReLU) else act(conv(segmap))   # ReLU in the conv's epilogue
        more = {"up2": True} if up2 else {}
        return spherenet.spade_norm_modulate(x, self.param_free_norm, actv, self.mlp_gamma, self.mlp_beta, slope, stats, **more)


class SPADEResnetBlock(nn.Module):
    """``architecture.py:20-62``."""

    def __init__(self, fin, fout, opt):
        super().__init__()
        self.learned_shortcut = fin != fout
        fmiddle = min(fin, fout)
        self.conv_0 = SphereConv2D(fin, fmiddle)
        self.conv_1 = SphereConv2D(fmiddle, fout)
        if self.learned_shortcut:
            self.conv_s = SphereConv2D(fin, fout)
        if "spectral" in opt.norm_G:
            sn = spherenet.fused_spectral_norm
            self.conv_0, self.conv_1 = sn(self.conv_0), sn(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = sn(self.conv_s)
        cfg = opt.norm_G.replace("spectral", "")
        self.norm_0 = SPADE(cfg, fin, opt.semantic_nc)
        self.norm_1 = SPADE(cfg, fmiddle, opt.semantic_nc)
        if self.learned_shortcut:
            self.norm_s = SPADE(cfg, fin, opt.semantic_nc)

    def forward(self, x, seg, out_slope=1.0, up2=False):
        """``out_slope``: a LeakyReLU the caller applies to the block's output (``generator.py:84`` before the last
        convolution), folded -- like the residual sum itself -- into ``conv_1``'s epilogue."""
        stats = None
        n0 = self.norm_0.param_free_norm
        ns = self.norm_s.param_free_norm if self.learned_shortcut else None
        # ``up2``: the caller's ``x = self.up(x)`` (generator.py:70-82) is left to this block.  With a learned shortcut x only
        # feeds norm_0 / norm_s, whose HIP kernels read the map before the upsample; otherwise it is upsampled here.
        fold = bool(up2 and self.learned_shortcut and x.is_cuda and x.shape[1] % 4 == 0 and isinstance(n0, nn.BatchNorm2d)
                    and isinstance(ns, nn.BatchNorm2d) and getattr(spherenet.spade_norm_modulate, "folds_upsample", False))
        if up2 and not fold:
            x = F.interpolate(x, scale_factor=2)
        if (self.learned_shortcut and isinstance(n0, nn.BatchNorm2d) and n0.training and ns.training and ns.eps == n0.eps
                and x.shape[1] % 4 == 0 and x.is_cuda):
            # training: norm_0 and norm_s normalise the same x with parameter-free BatchNorms -- one reduction serves both,
            # and each norm's running buffers are updated from it with its own momentum.  (In eval every norm uses its OWN
            # running statistics: a checkpoint may hold different buffers for the two.)
            stats = spherenet.spade_batch_stats(x, n0, also=(ns,), **({"repeat": 4} if fold else {}))
        x_s = self.conv_s(self.norm_s(x, seg, stats=stats, up2=fold)) if self.learned_shortcut else x
        dx = self.conv_0(self.norm_0(x, seg, slope=2e-1, stats=stats, up2=fold))   # leaky_relu(norm(.), 0.2) in the modulation
        return self.conv_1(self.norm_1(dx, seg, slope=2e-1), residual=x_s, act_slope=out_slope)   # act(x_s + dx)


def _norm_act(stage, x, slope):
    """``leaky_relu(stage(x), slope)`` for a ``nonspade_norm`` stage: Sequential(conv, InstanceNorm2d) runs its norm and the
    activation as one HIP launch; a bare convolution (norm 'none') just gets the activation."""
    mods = list(stage.children()) if isinstance(stage, nn.Sequential) else []
    if len(mods) == 2 and isinstance(mods[1], nn.InstanceNorm2d):
        return spherenet.instance_norm_act(mods[0](x), mods[1], slope)
    return F.leaky_relu(stage(x), slope)


class ConvEncoder(nn.Module):
    """crop -> 128x128 bilinear -> 5 stride-2 convs (+InstanceNorm) -> fc -> 32*ngf vector (``generator.py:90-125``)."""

    def __init__(self, opt):
        super().__init__()
        ndf = opt.ngf
        norm = nonspade_norm(opt.norm_E)
        self.layer1 = norm(nn.Conv2d(3, ndf, 3, stride=2, padding=1))
        self.layer2 = norm(nn.Conv2d(ndf, ndf * 2, 3, stride=2, padding=1))
        self.layer3 = norm(nn.Conv2d(ndf * 2, ndf * 4, 3, stride=2, padding=1))
        self.layer4 = norm(nn.Conv2d(ndf * 4, ndf * 8, 3, stride=2, padding=1))
        self.layer5 = norm(nn.Conv2d(ndf * 8, ndf * 8, 3, stride=2, padding=1))
        self.fc = nn.Linear(ndf * 8 * 4 * 4, 16 * ndf * 2 * 1)
        self.actvn = nn.LeakyReLU(0.2, False)

    def forward(self, x):
        x = F.interpolate(x, size=(128, 128), mode="bilinear")
        # every layer = conv -> InstanceNorm, each followed by the LeakyReLU (generator.py:113-122): norm + activation fused
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5):
            x = _norm_act(layer, x, self.actvn.negative_slope)
        return self.fc(x.reshape(x.size(0), -1))


class SPADEGenerator(nn.Module):
    """``forward(input, crop)``: the Gaussian map ``input`` (B,3,128,256) guides 7 SPADE blocks that upsample the
    encoded ``crop`` from (sh, sw) = (4, 8) to 128x256; output ``(tanh + 1) * 25`` (``generator.py:17-88``)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nf = opt.ngf
        self.sw, self.sh = self.compute_latent_vector_size(opt)
        self.head_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADEResnetBlock(16 * nf, 16 * nf, opt)
        self.up_0 = SPADEResnetBlock(16 * nf, 8 * nf, opt)
        self.up_1 = SPADEResnetBlock(8 * nf, 4 * nf, opt)
        self.up_2 = SPADEResnetBlock(4 * nf, 2 * nf, opt)
        self.up_3 = SPADEResnetBlock(2 * nf, 1 * nf, opt)
        self.up = nn.Upsample(scale_factor=2)
        self.sphere_conv1 = SphereConv2D(nf, 3, stride=1)
        self.netE = ConvEncoder(opt)

    @staticmethod
    def compute_latent_vector_size(opt):
        n_up = {"normal": 5, "more": 6, "most": 7}[opt.num_upsampling_layers]
        sw = opt.crop_size // (2 ** n_up)
        return sw, round(sw / opt.aspect_ratio)

    def forward(self, input, crop):
        spherenet.spectral_precompute(self)   # all 23 spectrally normalised weights of this pass in five launches
        guide = input
        x = self.netE(crop).view(-1, 16 * self.opt.ngf, 1, 2)
        x = F.interpolate(x, size=(self.sh, self.sw))
        x = self.head_0(x, guide)
        x = self.up(x)
        x = self.G_middle_0(x, guide)
        x = self.G_middle_1(x, guide)
        for blk in (self.up_0, self.up_1, self.up_2):
            x = blk(x, guide, up2=True)                     # = blk(self.up(x), guide): the upsample is folded into the block
        x = self.up_3(x, guide, out_slope=2e-1, up2=True)   # = F.leaky_relu(up_3(self.up(x), ...), 2e-1) of generator.py:84
        x = self.sphere_conv1(x)
        return (torch.tanh(x) + 1) * 25


class NLayerDiscriminator(nn.Module):
    """``discriminator.py:68-125``: SphereConv stages model0..model{n}, returns every stage's output."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        nf = opt.ndf
        norm = nonspade_norm(opt.norm_D)
        seq = [[SphereConv2D(opt.label_nc + opt.output_nc, nf, stride=2), nn.LeakyReLU(0.2, False)]]
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            seq += [[norm(SphereConv2D(nf_prev, nf, stride=stride)), nn.LeakyReLU(0.2, False)]]
        seq += [[SphereConv2D(nf, 3, stride=1)]]
        for n, s in enumerate(seq):
            self.add_module("model" + str(n), nn.Sequential(*s))

    def forward(self, input):
        results = [input]
        for sub in self.children():
            mods = list(sub.children())
            if len(mods) == 2 and isinstance(mods[1], nn.LeakyReLU) and isinstance(mods[0], nn.Sequential):
                results.append(_norm_act(mods[0], results[-1], mods[1].negative_slope))   # conv -> norm + LeakyReLU, one launch
            elif len(mods) == 2 and isinstance(mods[1], nn.LeakyReLU) and isinstance(mods[0], SphereConv2D):
                results.append(mods[0](results[-1], act_slope=mods[1].negative_slope))    # LeakyReLU in the conv's epilogue
            else:
                results.append(sub(results[-1]))
        return results[1:] if not self.opt.no_ganFeat_loss else results[-1]


class MultiscaleDiscriminator(nn.Module):
    """``discriminator.py:16-65``: ``num_D`` PatchGANs on an avg-pooled pyramid -> list[list[Tensor]]."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, NLayerDiscriminator(opt))

    @staticmethod
    def downsample(x):
        return F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, input):
        spherenet.spectral_precompute(self)
        result = []
        for _, D in self.named_children():
            out = D(input)
            result.append(out if not self.opt.no_ganFeat_loss else [out])
            input = self.downsample(input)
        return result


class GANLoss(nn.Module):
    """``loss.py:16-99`` (hinge default; ls / original / w kept)."""

    def __init__(self, gan_mode="hinge"):
        super().__init__()
        if gan_mode not in ("ls", "original", "w", "hinge"):
            raise ValueError("Unexpected gan_mode {}".format(gan_mode))
        self.gan_mode = gan_mode

    def loss(self, x, target_is_real, for_discriminator=True):
        if self.gan_mode == "original":
            return F.binary_cross_entropy_with_logits(x, torch.full_like(x, 1.0 if target_is_real else 0.0))
        if self.gan_mode == "ls":
            return F.mse_loss(x, torch.full_like(x, 1.0 if target_is_real else 0.0))
        if self.gan_mode == "hinge":
            if for_discriminator:
                v = (x - 1) if target_is_real else (-x - 1)
                return -torch.mean(torch.min(v, torch.zeros_like(v)))
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return -torch.mean(x)
        return -x.mean() if target_is_real else x.mean()

    def forward(self, x, target_is_real, for_discriminator=True):
        if isinstance(x, list):
            total = 0
            for pred in x:
                if isinstance(pred, list):
                    pred = pred[-1]
                lt = self.loss(pred, target_is_real, for_discriminator)
                bs = 1 if lt.dim() == 0 else lt.size(0)
                total = total + torch.mean(lt.view(bs, -1), dim=1)
            return total / len(x)
        return self.loss(x, target_is_real, for_discriminator)


def define_G(opt):
    return init_weights(SPADEGenerator(opt), opt.init_type, opt.init_variance)


def define_D(opt):
    return init_weights(MultiscaleDiscriminator(opt), opt.init_type, opt.init_variance)
