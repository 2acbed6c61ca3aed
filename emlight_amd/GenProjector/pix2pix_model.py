"""``Pix2PixModel`` -- generator / discriminator losses of the projector (reference ``models/pix2pix_model.py``).

``forward(data, mode)`` with ``mode in {'generator', 'discriminator', 'inference'}`` and the data dict keys
``input`` (Gaussian map, B,3,128,256), ``crop`` (B,3,128,128), ``warped`` (real HDR panorama), ``map`` (light mask).
Loss terms as ``pix2pix_model.py:92-141``: hinge GAN, mask-weighted feature matching (x50 off the lights),
cosine x5, VGG x5.  The VGG term (``vgg.py``) needs torchvision's pretrained VGG19 weights, which cannot be obtained
offline (parity of the VALUE unpinned): with ``opt.no_vgg_loss`` False the feature stack is built from
``opt.vgg_weights`` (a torchvision ``vgg19`` state dict) or, when absent, from seeded random weights -- the step then does
the reference's work.  A ``vgg_features`` callable may be passed instead.
"""
import torch
import torch.nn.functional as F

from . import networks


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

    def compute_discriminator_loss(self, inp, crop, real):
        with torch.no_grad():
            fake = self.generate_fake(inp, crop).detach()
        fake.requires_grad_()
        pred_fake, pred_real = self.discriminate(inp, fake, real)
        return {"D_Fake": self.criterionGAN(pred_fake, False, for_discriminator=True),
                "D_real": self.criterionGAN(pred_real, True, for_discriminator=True)}
