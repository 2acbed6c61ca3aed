"""Joint regression + projector training step (BASELINE configs[3]: "end-to-end regression+projector joint train").

The reference trains its two networks separately: ``RegressionNetwork/train.py`` fits the DenseNet to the
ground-truth light parameters, and ``GenProjector/train.py`` feeds the SPADE generator a Gaussian map rasterised
from the GROUND-TRUTH parameters inside its dataset (``GenProjector/data.py:86-102``).  The joint step defined here
closes that loop -- the generator is guided by the map of the PREDICTED parameters, and its losses reach the
encoder through the rasteriser's colour gradient (``eml_sg_rasterise_bwd_colors_f32``):

    pred   = DenseNet(crop)                                              RegressionNetwork/DenseNet.py:135-157
    L_reg  = 1000 EMD + 1000 MSE(dist) + .1 MSE(int) + 100 MSE(rgb) + MSE(amb)       train.py:90-98
    light  = dist (B,N,1) * (intensity * 5) * rgb (B,1,3)                 GenProjector/data.py:86-98
             (the regression target is intensity*alpha/500, the projector's lobe amplitude intensity*alpha*0.01)
    gmap   = convert_to_panorama(anchors, 0.0025, light) + ambient        data.py:99-102, util.py:222-245
             (the regression target ambient is already divided by 128*256 and carries alpha)
    G step: losses of pix2pix_model.py:92-127 on (gmap, crop128, warped, map); backward of L_reg + sum(G losses)
            updates the encoder (Adam 1e-4) and the generator (Adam, TTUR)          train.py:100-102 / model_trainer.py:34-42
    D step: pix2pix_model.py:129-141 on gmap.detach()                               model_trainer.py:44-50

One process per GPU; under torchrun the encoder, G and D are each wrapped in DistributedDataParallel (gradient
all-reduce over RCCL/xGMI) and SPADE's batch norm synchronises its statistics (``GenProjector/model_trainer.py``).

    python -m emlight_amd.joint --batch 8 --max_iters 10
    torchrun --nproc-per-node 8 -m emlight_amd.joint --batch 32
"""
import argparse

import torch
import torch.nn.functional as F

from .GenProjector import data as projector_data
from .GenProjector import networks
from .GenProjector.model_trainer import Trainer
from .RegressionNetwork.engine import RegressionTrainer, init_distributed, regression_loss


def predicted_gaussian_map(pred, ln, pano_hw=(128, 256)):
    """The generator's guide map from the encoder's outputs (module docstring): differentiable w.r.t. all four."""
    return projector_data.gaussian_map(pred["distribution"], pred["intensity"] * 500.0, pred["rgb_ratio"],
                                       pred["ambient"] * (pano_hw[0] * pano_hw[1]), ln=ln, pano_hw=pano_hw)


class JointTrainer:
    def __init__(self, opt=None, anchors=128, crop_hw=(240, 320), blur=.05, diameter=None, device="cuda", world=1,
                 pano_hw=(128, 256), encoder=None, sam_loss=None, sync_diameter=None, vgg_features=None):
        """``encoder`` / ``sam_loss``: see ``RegressionTrainer`` (CPU-only distributed tests inject the oracle's);
        ``vgg_features``: see ``GenProjector.model_trainer.Trainer``."""
        self.ln, self.pano_hw, self.world = anchors, tuple(pano_hw), world
        self.reg = RegressionTrainer(anchors=anchors, crop_hw=crop_hw, blur=blur, diameter=diameter, device=device,
                                     world=world, model=encoder, sam_loss=sam_loss, sync_diameter=sync_diameter)
        self.proj = Trainer(opt or networks.default_options(), device=device, world=world, vgg_features=vgg_features)
        self.losses = {}

    def projector_inputs(self, batch, pred):
        crop128 = F.interpolate(batch["crop"], size=(128, 128), mode="bilinear", align_corners=False)
        return {"input": predicted_gaussian_map(pred, self.ln, self.pano_hw), "crop": crop128,
                "warped": batch["warped"], "map": batch["map"]}

    def generator_step(self, batch):
        """Encoder + generator half of an iteration: one backward of L_reg + sum(G losses), Adam on both.  Returns the
        projector data dict (its guide map still attached to the encoder's graph) for ``discriminator_step``."""
        enc = self.reg.ddp if self.reg.ddp is not None else self.reg.model
        pm = self.proj.model
        pred = enc(batch["crop"])
        l_reg, terms = regression_loss(pred, batch, self.reg.sam_loss, self.ln)
        data = self.projector_inputs(batch, pred)
        g_losses, fake = pm(data, mode="generator")
        self.reg.optimizer.zero_grad(set_to_none=True)
        self.proj.optimizer_G.zero_grad()
        (l_reg + sum(g_losses.values()).mean()).backward()
        self.reg.reduce_gradients()
        self.proj.reduce_gradients("G")
        self.reg.optimizer.step()
        self.proj.optimizer_G.step()
        self.losses = {**terms, **g_losses}
        self.generated, self.guide = fake, data["input"]
        return data

    def discriminator_step(self, data):
        """Discriminator half on the same (detached) guide map, with the generator as ``generator_step`` left it."""
        pm = self.proj.model
        data_d = dict(data, input=data["input"].detach())
        self.proj.optimizer_D.zero_grad()
        d_losses = pm(data_d, mode="discriminator")
        sum(d_losses.values()).mean().backward()
        self.proj.reduce_gradients("D")
        self.proj.optimizer_D.step()
        self.losses = {**self.losses, **d_losses}
        return d_losses

    def step(self, batch):
        """One joint iteration on ``batch`` = regression keys (``crop, distribution, intensity, rgb_ratio, ambient``)
        + projector keys (``warped`` real HDR panorama (B,3,128,256), ``map`` light mask (B,1,128,256))."""
        self.discriminator_step(self.generator_step(batch))
        return self.losses


def joint_batch(batch, device, anchors=128, crop_hw=(240, 320), pano_hw=(128, 256), seed=1234):
    """Synthetic joint batch (SURVEY 8d recipes): regression GT + the projector's real panorama / light mask derived
    from the GROUND-TRUTH parameters (the guide map itself comes from the encoder at run time)."""
    from .RegressionNetwork.data import synthetic_batch
    p = synthetic_batch(batch, anchors, crop_hw, seed=seed, device=device)
    gt_map = predicted_gaussian_map(p, anchors, pano_hw)
    g = torch.Generator().manual_seed(seed + 7)
    noise = F.interpolate(torch.empty(batch, 1, 8, 16).uniform_(0.5, 1.5, generator=g), size=pano_hw, mode="bilinear",
                          align_corners=False).to(device)
    warped = gt_map * noise
    luma = 0.3 * warped[:, 0] + 0.59 * warped[:, 1] + 0.11 * warped[:, 2]
    p["warped"] = warped
    p["map"] = (luma > 0.05 * luma.amax(dim=(1, 2), keepdim=True)).float().unsqueeze(1)
    return p


def main(argv=None):
    from emlight_amd import _runtime
    _runtime.entry_point_defaults()   # kernel arguments in device memory, recorded library-GEMM selection: an entry point's choice
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[3]: 256 over 8 GPUs)")
    ap.add_argument("--anchors", type=int, default=128)
    ap.add_argument("--crop_hw", type=int, nargs=2, default=(240, 320))
    ap.add_argument("--blur", type=float, default=.05)
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--ndf", type=int, default=64)
    ap.add_argument("--max_iters", type=int, default=100)
    networks.add_vgg_arguments(ap)
    args = ap.parse_args(argv)
    rank, local, world = init_distributed()
    dev = "cuda:%d" % local
    tr = JointTrainer(networks.default_options(ngf=args.ngf, ndf=args.ndf, **networks.vgg_options(args, verbose=rank == 0)),
                      anchors=args.anchors,
                      crop_hw=tuple(args.crop_hw), blur=args.blur, device=dev, world=world)
    for it in range(args.max_iters):
        batch = joint_batch(args.batch, dev, args.anchors, tuple(args.crop_hw), seed=1234 + rank + 977 * it)
        losses = tr.step(batch)
        if rank == 0 and it % 10 == 0:   # the only host syncs
            print("iter %d " % it + " ".join("%s: %.4f" % (k, float(v.mean())) for k, v in losses.items()))


if __name__ == "__main__":
    main()
