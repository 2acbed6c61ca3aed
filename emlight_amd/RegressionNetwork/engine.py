"""One regression training step (the hot loop of ``RegressionNetwork/train.py:79-102``),
shared by ``train.py`` and ``bench.py``: forward, weighted loss, backward, Adam -- one
process per GPU, gradients all-reduced by DDP (RCCL over xGMI) when world_size > 1."""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .DenseNet import DenseNet
from .geomloss import SamplesLoss


def regression_loss(pred, gt, sam_loss, ln):
    """``train.py:90-98``: 1000*sum(EMD) + 1000*MSE(dist) + .1*MSE(int) + 100*MSE(rgb) + MSE(amb)."""
    dist_pred = pred["distribution"].view(-1, ln, 1)
    dist_gt = gt["distribution"].view(-1, ln, 1)
    terms = {
        "dist_emloss": sam_loss(dist_pred, dist_gt).sum() * 1000.0,
        "dist_l2loss": F.mse_loss(dist_pred, dist_gt) * 1000.0,
        "intensity_loss": F.mse_loss(pred["intensity"], gt["intensity"]) * 0.1,
        "rgb_loss": F.mse_loss(pred["rgb_ratio"], gt["rgb_ratio"]) * 100.0,
        "ambient_loss": F.mse_loss(pred["ambient"], gt["ambient"]) * 1.0,
    }
    total = (terms["dist_emloss"] + terms["dist_l2loss"] + terms["intensity_loss"]
             + terms["rgb_loss"] + terms["ambient_loss"])
    return total, terms


def init_distributed():
    """torchrun env -> (rank, local_rank, world).  backend 'nccl' is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and local >= torch.cuda.device_count():
        # More ranks than GPUs.  RCCL wants one rank per device, so this is an error for a real launch; the 2-rank tests
        # on a 1-GPU box opt in (gloo transport) and share the device.
        if os.environ.get("EML_DIST_BACKEND") != "gloo" and os.environ.get("EML_SHARE_GPUS") != "1":
            raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible: one process per GPU (set EML_DIST_BACKEND=gloo "
                               "or EML_SHARE_GPUS=1 to let ranks share a device in tests)" % (local, torch.cuda.device_count()))
        local %= torch.cuda.device_count()
    from .._dist import single_rank_dry_run
    if (world > 1 or single_rank_dry_run()) and not dist.is_initialized():
        # 'nccl' is RCCL on ROCm; EML_DIST_BACKEND=gloo lets two ranks share one GPU in tests.  EML_DIST_SINGLE=1: the
        # one-rank dry run of the multi-GPU path (_dist.py) -- the group is initialised with a single rank as well
        backend = os.environ.get("EML_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if world == 1 and "MASTER_ADDR" not in os.environ:   # no launcher: rendezvous with ourselves on a free local port
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
            s.close()
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


class RegressionTrainer:
    """Model + Sinkhorn criterion + Adam(1e-4, (.9,.999)) (``train.py:55-61``)."""

    def __init__(self, anchors=96, crop_hw=(192, 256), blur=.025, diameter=None, lr=1e-4,
                 betas=(0.9, 0.999), device="cuda", world=1, bucket_cap_mb=64, model=None, sam_loss=None,
                 sync_diameter=None):
        """``model`` / ``sam_loss``: pre-built modules to train instead of the HIP ``DenseNet`` / ``SamplesLoss``
        (the CPU-only distributed tests inject the oracle's stock-op restatements; the product never does).
        ``sync_diameter``: derive the Sinkhorn eps-schedule from the range of the GLOBAL batch (2-float all-reduce per
        step) so that N ranks x B reproduce the single-process run with N*B samples (sinkhorn_divergence.py:9-18);
        default: on when world > 1 and no fixed ``diameter`` is given."""
        from .._dist import dp_wrap, own_buckets, GradientBuckets
        if sync_diameter is None:
            sync_diameter = dp_wrap(world) and diameter is None
        self.ln = anchors
        self.device = torch.device(device)
        self.model = (DenseNet(anchors=anchors, crop_hw=crop_hw) if model is None else model).to(self.device)
        self.model.train()
        self.sam_loss = sam_loss or SamplesLoss("sinkhorn", p=2, blur=blur, diameter=diameter, anchors=anchors,
                                                sync_diameter=sync_diameter)
        self.ddp = self.buckets = None
        if dp_wrap(world) and own_buckets():
            # 37.3 MB of f32 gradients: the encoder's backward is ONE autograd node, so its one bucket is packed and reduced
            # when that node returns (GradientBuckets: no per-parameter copies)
            self.buckets = GradientBuckets(self.model.parameters(), world, cap_mb=bucket_cap_mb, name="encoder",
                                           buffers=list(self.model.buffers()))
        elif dp_wrap(world):
            # DenseNet BN stays per-rank (plain nn.BatchNorm2d in the reference); only the
            # 37.3 MB of f32 gradients cross xGMI, in one bucket overlapped with backward.
            self.ddp = torch.nn.parallel.DistributedDataParallel(
                self.model, device_ids=[self.device.index] if self.device.type == "cuda" else None,
                bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, broadcast_buffers=False)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, betas=betas,
                                          fused=self.device.type == "cuda")   # one launch instead of the foreach passes

    def reduce_gradients(self):
        """Between backward and the optimizer step: the gradients' all-reduce is complete (DDP: its hooks have done it)."""
        if self.buckets is not None:
            self.buckets.finish()

    def step(self, batch):
        net = self.ddp if self.ddp is not None else self.model
        pred = net(batch["crop"])
        loss, terms = regression_loss(pred, batch, self.sam_loss, self.ln)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.reduce_gradients()
        self.optimizer.step()
        self.last_pred = pred   # train.py visualises the training batch's own prediction (train.py:110-145)
        return loss, terms
