"""GPU: the N>1 path with the HIP engine -- two ranks (sharing the one GPU of the test box, gloo transport so
RCCL's one-rank-per-device rule does not apply) run the real RegressionTrainer under DDP."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world), EML_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    from emlight_amd.RegressionNetwork.engine import RegressionTrainer, init_distributed
    r, local, w = init_distributed()
    torch.manual_seed(0)
    tr = RegressionTrainer(anchors=32, crop_hw=(64, 96), blur=.05, device="cuda:0", world=w)
    batch = synthetic_batch(2, 32, (64, 96), seed=1234 + rank, device="cuda:0")
    losses = []
    for _ in range(3):
        loss, _ = tr.step(batch)
        losses.append(float(loss))
    flat = torch.cat([q.detach().reshape(-1) for q in tr.model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(w)]
    dist.all_gather(gathered, flat)
    np.save(os.path.join(out_dir, "r%d.npy" % rank),
            np.array([float((gathered[0] - gathered[1]).abs().max()), float(np.isfinite(losses).all()), losses[0], losses[-1]]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_hip_engine_ddp(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert r0[0] == 0.0 and r1[0] == 0.0, "replicas diverged under DDP"
    assert r0[1] == 1.0 and r1[1] == 1.0


def _projector_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world), EML_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.GenProjector import networks
    from emlight_amd.GenProjector.data import projector_batch
    from emlight_amd.GenProjector.model_trainer import Trainer
    r, local, w = init_distributed()
    torch.manual_seed(0)
    tr = Trainer(networks.default_options(ngf=4, ndf=4), device="cuda:0", world=w)   # DDP(G), DDP(D); SPADE syncs its BN sums
    data = projector_batch(1, "cuda:0", seed=50 + rank)
    tr.step(data)
    ok = all(bool(torch.isfinite(v).all()) for v in tr.get_latest_losses().values())
    diffs = []
    for net in (tr.model.netG, tr.model.netD):
        flat = torch.cat([q.detach().reshape(-1) for q in net.parameters()] +
                         [b.detach().float().reshape(-1) for b in net.buffers()])   # incl. SyncBN running statistics
        gathered = [torch.empty_like(flat) for _ in range(w)]
        dist.all_gather(gathered, flat)
        diffs.append(float((gathered[0] - gathered[1]).abs().max()))
    np.save(os.path.join(out_dir, "p%d.npy" % rank), np.array([float(ok)] + diffs))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_projector_syncbn_hip_sphereconv(tmp_path):
    """Projector trainer on two ranks: HIP SphereConv2D / SPADE kernels under DDP with SyncBatchNorm (the job of the
    reference's vendored sync_batchnorm): parameters AND running statistics of both networks stay identical."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_projector_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert p0[0] == 1.0 and p1[0] == 1.0
    assert p0[1] == 0.0 and p0[2] == 0.0, "projector replicas diverged: %s" % p0
