"""``Pix2PixModel`` -- generator / discriminator losses of the projector (reference ``models/pix2pix_model.py``).

``forward(data, mode)`` with ``mode in {'generator', 'discriminator', 'inference'}`` and the data dict keys
``input`` (Gaussian map, B,3,128,256), ``crop`` (B,3,128,128), ``warped`` (real HDR panorama), ``map`` (light mask).
Loss terms as ``pix2pix_model.py:92-141``: hinge GAN, mask-weighted feature matching (x50 off the lights),
cosine x5, VGG x5.  The VGG term (``vgg.py``) needs torchvision's pretrained VGG19 weights, which cannot be obtained
offline (parity of the VALUE unpinned): with ``opt.no_vgg_loss`` False the feature stack is built from
``opt.vgg_weights`` (a torchvision ``vgg19`` state dict) or, when absent, from seeded random weights -- the step then does
the reference's work.  A ``vgg_features`` callable may be passed instead.
"""
import warnings

import torch
import torch.nn.functional as F

from .._knobs import knob_flag
from . import networks


class _NoGradGraph:
    """The discriminator step's generator pass (``pix2pix_model.py:124-126``: ``with torch.no_grad(): fake = G(...)``) as a
    captured HIP graph.  That pass is ~250 launches, a third of them 5-20 us kernels of the low-resolution blocks behind 30-50 us
    of Python each: the GPU idles for the host there (``profiles/r05_gaps_joint.txt``).  It has no autograd state and fixed
    shapes, and everything it reads or updates is updated IN PLACE between replays (parameters by Adam, the power iteration's
    ``weight_u`` / ``weight_v``, BatchNorm's running statistics), so one capture replays correctly for the rest of the run: a
    replay is the same kernels on the same addresses as an eager call (``test_discriminator_step_graph_equals_eager``).
    Round 5's attempt died on ``hipErrorStreamCaptureUnsupported``; ``tools/capture_probe.py`` located the one offender (a
    Python scalar stored into a device tensor = a host-to-device copy, ``spherenet._reduce_sums``), now a fill kernel.
    Captured on the second call per (shapes, mode) -- the first runs eagerly so that every per-geometry table, LDS attribute
    and library handle exists; any failure to capture switches the instance back to eager calls for good, loudly."""

    def __init__(self, net):
        self.net, self.graph, self.key, self.failed, self.calls = net, None, None, False, 0
        self.s_inp = self.s_crop = self.out = None

    def __deepcopy__(self, memo):
        return None   # a copy of the model captures its own graph

    def __reduce__(self):
        return (type(None), ())   # nor does a pickled model carry one

    def __call__(self, inp, crop):
        if self.failed:
            return self.net(inp, crop)
        key = (tuple(inp.shape), tuple(crop.shape), inp.device, inp.dtype, crop.dtype, bool(self.net.training))
        if key != self.key:
            self.graph, self.out, self.key, self.calls = None, None, key, 0
        self.calls += 1
        if self.calls == 1:
            return self.net(inp, crop)
        if self.graph is None:
            try:
                self._capture(inp, crop)
            except Exception as e:   # noqa: BLE001 -- whatever the runtime refuses: the eager pass is always available
                self.failed, self.graph, self.out = True, None, None
                torch.cuda.synchronize(inp.device)
                warnings.warn("the discriminator step's generator pass could not be captured as a graph (%s: %s); running it "
                              "eagerly from now on" % (type(e).__name__, str(e).splitlines()[0][:200]))
                return self.net(inp, crop)
        self.s_inp.copy_(inp)
        self.s_crop.copy_(crop)
        self.graph.replay()
        return self.out.clone()   # the static output is overwritten by the next replay

    def _capture(self, inp, crop):
        self.s_inp, self.s_crop = inp.detach().clone(), crop.detach().clone()
        dev = inp.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # the libraries' per-stream state (BLAS handle, workspace) outside the capture
            a = torch.ones(64, 64, device=dev)
            torch.mm(a, a)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = self.net(self.s_inp, self.s_crop)
        self.graph, self.out = graph, out


class Pix2PixModel(torch.nn.Module):
    def __init__(self, opt, vgg_features=None):
        super().__init__()
        self.opt = opt
        self.netG = networks.define_G(opt)
        self.netD = networks.define_D(opt) if opt.isTrain else None
        if opt.isTrain:
            self.criterionGAN = networks.GANLoss(opt.gan_mode)
            self.criterionFeat = torch.nn.L1Loss()
            if not opt.no_vgg_loss and vgg_features is None:
                # the reference always adds the term (pix2pix_model.py:119-120); torchvision's ImageNet weights cannot be
                # obtained offline, so they are injectable: opt.vgg_weights = path of a torchvision vgg19 state dict,
                # otherwise a seeded random stack that does the same work (vgg.py)
                from .vgg import VGG19Features
                path = getattr(opt, "vgg_weights", None)
                if not path and not getattr(opt, "vgg_random", False):
                    # never optimise against random features silently
                    raise ValueError("the VGG perceptual term needs opt.vgg_weights (a torchvision vgg19 state dict: the "
                                     "reference's objective) or an explicit opt.vgg_random=True (seeded random features: "
                                     "the reference's WORK, for timing -- not its loss); or set opt.no_vgg_loss=True")
                vgg_features = VGG19Features(torch.load(path, map_location="cpu") if path else None)
        self.vgg_features = vgg_features
        # which perceptual term this model is trained against (saved next to the checkpoints by the entry points)
        self.vgg_variant = "off" if (not opt.isTrain or opt.no_vgg_loss) else getattr(vgg_features, "variant", "injected")
        # D as it is called in the discriminator step (the trainer swaps in its DistributedDataParallel wrapper); kept out of
        # the module tree so that state_dict keys stay the reference's
        object.__setattr__(self, "netD_train", self.netD)

    def forward(self, data, mode):
        dev = next(self.netG.parameters()).device
        inp, crop, real, mask = (data[k].to(dev) for k in ("input", "crop", "warped", "map"))
        if mode == "generator":
            return self.compute_generator_loss(inp, crop, real, mask)
        if mode == "discriminator":
            return self.compute_discriminator_loss(inp, crop, real)
        if mode == "inference":
            with torch.no_grad():
                return self.generate_fake(inp, crop)
        raise ValueError("|mode| is invalid")

    def create_optimizers(self, opt):
        G_lr, D_lr = (opt.lr, opt.lr) if opt.no_TTUR else (opt.lr / 2, opt.lr * 2)
        # same update rule as the reference's torch.optim.Adam; on the GPU as ONE fused launch per optimizer instead of the
        # ~10 foreach passes over G's 385 MB of parameters and moments
        fused = all(q.is_cuda for q in self.parameters())
        return (torch.optim.Adam(self.netG.parameters(), lr=G_lr, betas=(opt.beta1, opt.beta2), fused=fused),
                torch.optim.Adam(self.netD.parameters(), lr=D_lr, betas=(opt.beta1, opt.beta2), fused=fused))

    def generate_fake(self, inp, crop):
        return self.netG(inp, crop)

    def discriminate_raw(self, inp, fake, real, for_generator=False):
        """D on cat(fake pair, real pair) along the batch: list (per discriminator) of lists (per stage) of (2B, ...) maps.
        ``for_generator``: the generator step uses D as a fixed critic.  The reference's backward also fills D's ``.grad``
        there, but nothing ever reads it (``optimizer_D.zero_grad()`` opens the D step, model_trainer.py:44-46): D's parameters
        are held out of that graph -- no weight gradients, no spectral-norm backward, and under DDP no all-reduce of them."""
        both = torch.cat([torch.cat([inp, fake], dim=1), torch.cat([inp, real], dim=1)], dim=0)
        if for_generator:
            held = [q for q in self.netD.parameters() if q.requires_grad]
            for q in held:
                q.requires_grad_(False)
            try:
                return self.netD(both)
            finally:
                for q in held:
                    q.requires_grad_(True)
        return self.netD_train(both)

    def discriminate(self, inp, fake, real, for_generator=False):
        out = self.discriminate_raw(inp, fake, real, for_generator)
        fake_p = [[t[:t.size(0) // 2] for t in p] for p in out]
        real_p = [[t[t.size(0) // 2:] for t in p] for p in out]
        return fake_p, real_p

    def compute_generator_loss(self, inp, crop, real, mask):
        losses = {}
        fake = self.generate_fake(inp, crop)
        out = self.discriminate_raw(inp, fake, real, for_generator=True)
        pred_fake = [[t[:t.size(0) // 2] for t in p] for p in out]
        losses["GAN"] = self.criterionGAN(pred_fake, True, for_discriminator=False)
        if not self.opt.no_ganFeat_loss:
            num_D = len(out)
            feats, masks = [], []
            for i in range(num_D):
                for j in range(len(out[i]) - 1):
                    mask = F.interpolate(mask, size=out[i][j].shape[2:])   # the reference re-interpolates the running mask
                    feats.append(out[i][j])
                    masks.append(mask)
            # reference: L1(f*m + f*(1-m)*50, r*m + r*(1-m)*50) -- both sides carry the same per-pixel weight
            # m + 50(1-m) = 50 - 49m, so a term is mean(|(f - r) * (50 - 49m)|); differs from the literal form by f32 rounding only
            from . import l1_terms
            if fake.is_cuda and l1_terms.ENABLED and feats:   # (no intermediate maps, --n_layers_D 1: the zero below)
                # all (discriminator, stage) terms in one launch each way, straight on the maps of cat(fake, real)
                losses["GAN_Feat"] = l1_terms.feature_matching(feats, masks, num_D).reshape(1)
            else:
                feat = fake.new_zeros(1)
                for t, m in zip(feats, masks):
                    half = t.size(0) // 2
                    feat = feat + ((t[:half] - t[half:].detach()) * (50.0 - 49.0 * m)).abs().mean() / num_D
                losses["GAN_Feat"] = feat
        if not self.opt.no_vgg_loss:
            from .vgg import vgg_loss
            losses["VGG"] = vgg_loss(self.vgg_features, fake, real) * 5
        cos = torch.nn.CosineSimilarity(dim=1, eps=1e-20)
        losses["COS"] = (1 - cos(fake, real)).mean() * 5
        return losses, fake

    # EML_GRAPH_DSTEP=0: A/B knob -- the discriminator step's generator pass eagerly, as in rounds 1-5
    graph_dstep = knob_flag("EML_GRAPH_DSTEP", True)

    def _fake_for_discriminator(self, inp, crop):
        """``generate_fake`` under no_grad; as a replayed graph where that is possible: one process per job (with more ranks the
        pass holds SPADE's BatchNorm all-reduces and goes through the DDP wrapper: eager), a GPU generator in training mode,
        shapes that repeat."""
        from . import spherenet
        if (Pix2PixModel.graph_dstep and inp.is_cuda and "generate_fake" not in self.__dict__ and not spherenet._bn_sync()
                and not torch.cuda.is_current_stream_capturing()):
            g = self.__dict__.get("_dstep_graph")
            if g is None:
                g = _NoGradGraph(self.netG)
                object.__setattr__(self, "_dstep_graph", g)
            return g(inp, crop)
        return self.generate_fake(inp, crop).detach()

    def compute_discriminator_loss(self, inp, crop, real):
        with torch.no_grad():
            fake = self._fake_for_discriminator(inp, crop)
        fake.requires_grad_()
        pred_fake, pred_real = self.discriminate(inp, fake, real)
        return {"D_Fake": self.criterionGAN(pred_fake, False, for_discriminator=True),
                "D_real": self.criterionGAN(pred_real, True, for_discriminator=True)}
