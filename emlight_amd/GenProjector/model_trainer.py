"""``Trainer`` of the projector (reference ``GenProjector/model_trainer.py``): one G step + one D step.

The reference wraps the model in a single-process multi-thread ``DataParallelWithCallback``; here each GPU is
its own process: G's and D's gradients are all-reduced over RCCL/xGMI in 25 MB buckets overlapped with backward
(``_dist.GradientBuckets``; ``EML_DP_BUCKETS=0``: DistributedDataParallel wrappers); SPADE's param-free BatchNorm all-reduces its (2C+1)-float sums
itself (``spherenet.spade_batch_stats`` / the modulation's backward), which is what the vendored ``sync_batchnorm``
package did."""
import os

import torch

from .pix2pix_model import Pix2PixModel


class Trainer:
    def __init__(self, opt, device="cuda", world=1, vgg_features=None):
        """``vgg_features``: a ready VGG19 feature callable for the perceptual term (see ``Pix2PixModel``), instead of
        building one from ``opt.vgg_weights``."""
        self.opt = opt
        self.model = Pix2PixModel(opt, vgg_features=vgg_features).to(device)
        self.world = world
        from .._dist import dp_wrap, own_buckets, GradientBuckets
        self.bucketsG = self.bucketsD = None
        if dp_wrap(world) and own_buckets():
            # the package's own reducer: a bucket is packed by one multi-tensor copy and all-reduced while backward continues
            self.bucketsG = GradientBuckets(self.model.netG.parameters(), world, cap_mb=25, name="generator",
                                            buffers=list(self.model.netG.buffers()))
            self.bucketsD = GradientBuckets(self.model.netD.parameters(), world, cap_mb=25, name="discriminator",
                                            buffers=list(self.model.netD.buffers()))
        elif dp_wrap(world):
            ids = [torch.device(device).index] if str(device).startswith("cuda") else None
            # gradients live IN the all-reduce buckets (views): no copy of the reduced 473 MB back into .grad
            kw = dict(device_ids=ids, bucket_cap_mb=25, gradient_as_bucket_view=True)
            self._ddpG = torch.nn.parallel.DistributedDataParallel(self.model.netG, **kw)
            self._ddpD = torch.nn.parallel.DistributedDataParallel(self.model.netD, **kw)
            # the model calls generate_fake / the discriminator step's D: route those through the DDP wrappers (the
            # generator step calls the bare D with its parameters held out of the graph: nothing to reduce)
            self.model.generate_fake = lambda inp, crop: self._ddpG(inp, crop)
            object.__setattr__(self.model, "netD_train", self._ddpD)
        self.optimizer_G, self.optimizer_D = self.model.create_optimizers(opt)
        self.old_lr = opt.lr
        self.g_losses, self.d_losses, self.generated = {}, {}, None

    def reduce_gradients(self, which):
        b = self.bucketsG if which == "G" else self.bucketsD
        if b is not None:
            b.finish()

    def run_generator_one_step(self, data):
        self.optimizer_G.zero_grad()
        g_losses, generated = self.model(data, mode="generator")
        sum(g_losses.values()).mean().backward()
        self.reduce_gradients("G")
        self.optimizer_G.step()
        self.g_losses, self.generated = g_losses, generated

    def run_discriminator_one_step(self, data):
        self.optimizer_D.zero_grad()
        d_losses = self.model(data, mode="discriminator")
        sum(d_losses.values()).mean().backward()
        self.reduce_gradients("D")
        self.optimizer_D.step()
        self.d_losses = d_losses

    def step(self, data):
        """One training iteration as ``GenProjector/train.py:33-37``."""
        self.run_generator_one_step(data)
        self.run_discriminator_one_step(data)

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def save(self, epoch, save_dir):
        """``<epoch>_net_G.pth`` / ``<epoch>_net_D.pth`` (reference ``util.py:173-178``, ``pix2pix_model.py:66-70``)."""
        os.makedirs(save_dir, exist_ok=True)
        for label, net in (("G", self.model.netG), ("D", self.model.netD)):
            torch.save(net.state_dict(), os.path.join(save_dir, "%s_net_%s.pth" % (epoch, label)))

    def load(self, epoch, save_dir):
        """Resume: ``--continue_train --which_epoch <epoch>`` (reference ``util.py:181-191``, ``pix2pix_model.py:80-88``)."""
        dev = next(self.model.netG.parameters()).device
        for label, net in (("G", self.model.netG), ("D", self.model.netD)):
            net.load_state_dict(torch.load(os.path.join(save_dir, "%s_net_%s.pth" % (epoch, label)), map_location=dev))

    def update_learning_rate(self, epoch, niter=50, niter_decay=0):
        """Linear decay after ``niter`` epochs with TTUR (``model_trainer.py:68-88``)."""
        new_lr = self.old_lr - self.opt.lr / niter_decay if (epoch > niter and niter_decay > 0) else self.old_lr
        if new_lr != self.old_lr:
            g, d = (new_lr, new_lr) if self.opt.no_TTUR else (new_lr / 2, new_lr * 2)
            for pg in self.optimizer_G.param_groups:
                pg["lr"] = g
            for pg in self.optimizer_D.param_groups:
                pg["lr"] = d
            self.old_lr = new_lr
