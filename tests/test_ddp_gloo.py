"""CPU, world_size 2, gloo: the N>1 path of the regression trainer and of bench.py's timing.

The batch dimension shards (one process per GPU, independent samples); the only collective on the
data path is DDP's gradient all-reduce (RCCL on the GPU box, gloo here).  The HIP kernels need a GPU,
so the ranks run the stock-op engine and an oracle-backed Sinkhorn term -- what is under test is the
distributed plumbing: rank env handling, DDP wrapping, gradient averaging, identical replicas after a
step, and the max-over-ranks timing contract."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, buckets="1"):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["EML_DP_BUCKETS"] = buckets   # "1": the package's GradientBuckets; "0": torch's DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import oracle
    import bench
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    from emlight_amd.RegressionNetwork.engine import RegressionTrainer, init_distributed, regression_loss
    r, local, w = init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    anchors, crop = 16, (32, 32)
    torch.manual_seed(0)  # identical initial replicas, like loading one checkpoint on every rank
    M = oracle.anchor_cost_matrix(anchors)
    # CPU ranks: the oracle's stock-op encoder and Sinkhorn term are injected (the product's are HIP-only)
    tr = RegressionTrainer(anchors=anchors, crop_hw=crop, blur=.05, device="cpu", world=w,
                           model=oracle.OracleDenseNet(anchors=anchors, crop_hw=crop),
                           sam_loss=lambda x, y: oracle.samples_loss(x, y, M, blur=.05))
    batch = synthetic_batch(2, anchors, crop, seed=1234 + rank)  # each rank its own shard

    # reference: local gradients on a private copy, averaged by hand over the ranks
    import copy
    priv = copy.deepcopy(tr.model)
    loss, _ = regression_loss(priv(batch["crop"]), batch, tr.sam_loss, anchors)
    loss.backward()
    want = []
    for q in priv.parameters():
        g = q.grad.clone()
        dist.all_reduce(g)
        want.append(g / w)

    # forward / backward through the data-parallel path (DDP's wrapper or the model with GradientBuckets' hooks), then compare
    assert (tr.ddp is None) == (buckets == "1") and (tr.buckets is not None) == (buckets == "1")
    pred = (tr.ddp if tr.ddp is not None else tr.model)(batch["crop"])
    loss2, _ = regression_loss(pred, batch, tr.sam_loss, anchors)
    tr.optimizer.zero_grad(set_to_none=True)
    loss2.backward()
    tr.reduce_gradients()
    worst = 0.0
    for q, g in zip(tr.model.parameters(), want):
        worst = max(worst, float((q.grad - g).abs().max() / (g.abs().max() + 1e-12)))
    tr.optimizer.step()
    # replicas stay identical after the step
    flat = torch.cat([q.detach().reshape(-1) for q in tr.model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(w)]
    dist.all_gather(gathered, flat)
    replica_diff = float((gathered[0] - gathered[1]).abs().max())

    # bench.py timing contract: every rank gets the MAX over ranks
    dt = bench.run_timed(lambda: time.sleep(0.05 * (rank + 1)), steps=2, warmup=1, world=w, device="cpu")
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([worst, replica_diff, dt]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("buckets", ["1", "0"])
def test_two_rank_gloo_training_step(tmp_path, buckets):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), buckets), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert r0[0] < 1e-5 and r1[0] < 1e-5, "DDP gradient != mean of per-rank gradients: %s %s" % (r0, r1)
    assert r0[1] == 0.0 and r1[1] == 0.0, "replicas diverged after one step"
    assert abs(r0[2] - r1[2]) < 1e-9 and r0[2] >= 0.2 - 1e-3, "timing must be the max over ranks: %s %s" % (r0, r1)


def _projector_worker(rank, world, port, out_dir, buckets="1"):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["EML_DP_BUCKETS"] = buckets   # "1": the package's GradientBuckets; "0": torch's DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.GenProjector import networks
    from emlight_amd.GenProjector.model_trainer import Trainer
    import oracle
    r, local, w = init_distributed()
    torch.manual_seed(0)  # identical initial replicas
    with oracle.stock_sphere_ops():  # CPU ranks: the reference's stock ops (the HIP SphereConv2D has no CPU path)
        tr = Trainer(networks.default_options(ngf=2, ndf=2), device="cpu", world=w)
        g = torch.Generator().manual_seed(100 + rank)  # each rank its own shard
        data = {"input": torch.rand(1, 3, 128, 256, generator=g) * 5, "crop": torch.rand(1, 3, 128, 128, generator=g),
                "warped": torch.rand(1, 3, 128, 256, generator=g) * 5,
                "map": (torch.rand(1, 1, 128, 256, generator=g) > 0.5).float()}
        tr.step(data)
    ok = all(bool(torch.isfinite(v).all()) for v in tr.get_latest_losses().values())
    diffs = []
    for net in (tr.model.netG, tr.model.netD):  # DDP all-reduced gradients => replicas identical after G and D steps
        flat = torch.cat([q.detach().reshape(-1) for q in net.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(w)]
        dist.all_gather(gathered, flat)
        diffs.append(float((gathered[0] - gathered[1]).abs().max()))
    np.save(os.path.join(out_dir, "p%d.npy" % rank), np.array([float(ok)] + diffs))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("buckets", ["1", "0"])
def test_two_rank_gloo_projector_step(tmp_path, buckets):
    """GenProjector trainer under DDP (G and D wrapped separately, model_trainer.py): one G step + one D step on two
    ranks with different shards leaves both replicas of both networks identical."""
    port = _free_port()
    mp.spawn(_projector_worker, args=(2, port, str(tmp_path), buckets), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert p0[0] == 1.0 and p1[0] == 1.0, "non-finite projector losses"
    assert p0[1] == 0.0 and p0[2] == 0.0, "projector replicas diverged after one DDP step: %s" % p0


def _joint_worker(rank, world, port, out_dir, buckets="1"):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["EML_DP_BUCKETS"] = buckets   # "1": the package's GradientBuckets; "0": torch's DistributedDataParallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import oracle
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.GenProjector import networks
    from emlight_amd.joint import JointTrainer, joint_batch
    r, local, w = init_distributed()
    torch.manual_seed(0)  # identical initial replicas
    anchors, crop = 16, (32, 32)
    M = oracle.anchor_cost_matrix(anchors)
    # CPU ranks: oracle encoder / Sinkhorn / rasteriser and stock-op SphereNet ops are injected (the product's are HIP-only)
    with oracle.stock_sphere_ops(), oracle.stock_rasteriser():
        tr = JointTrainer(networks.default_options(ngf=2, ndf=2), anchors=anchors, crop_hw=crop, device="cpu", world=w,
                          encoder=oracle.OracleDenseNet(anchors=anchors, crop_hw=crop),
                          sam_loss=lambda x, y: oracle.samples_loss(x, y, M, blur=.05))
        batch = joint_batch(1, "cpu", anchors, crop, seed=300 + rank)  # each rank its own shard
        losses = tr.step(batch)
    ok = all(bool(torch.isfinite(v).all()) for v in losses.values())
    enc_grad = float(sum(q.grad.abs().sum() for q in tr.reg.model.parameters()))
    diffs = []
    for net in (tr.reg.model, tr.proj.model.netG, tr.proj.model.netD):
        flat = torch.cat([q.detach().reshape(-1) for q in net.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(w)]
        dist.all_gather(gathered, flat)
        diffs.append(float((gathered[0] - gathered[1]).abs().max()))
    np.save(os.path.join(out_dir, "j%d.npy" % rank), np.array([float(ok), enc_grad] + diffs))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("buckets", ["1", "0"])
def test_two_rank_gloo_joint_step(tmp_path, buckets):
    """Joint regression+projector trainer (BASELINE configs[3]) under DDP on two ranks with different shards: the
    encoder, generator and discriminator replicas are identical after the iteration (three DDP-wrapped networks,
    the encoder's and the generator's gradients all-reduced inside ONE backward)."""
    port = _free_port()
    mp.spawn(_joint_worker, args=(2, port, str(tmp_path), buckets), nprocs=2, join=True)
    j0, j1 = np.load(tmp_path / "j0.npy"), np.load(tmp_path / "j1.npy")
    assert j0[0] == 1.0 and j1[0] == 1.0, "non-finite joint losses"
    assert j0[1] > 0 and j0[1] == j1[1], "DDP-averaged encoder gradients must be identical and non-zero on both ranks"
    assert j0[2] == 0.0 and j0[3] == 0.0 and j0[4] == 0.0, "joint replicas diverged after one DDP step: %s" % j0


def test_bench_never_reports_a_wrong_gpu_count():
    """`bench.py --gpus 2` must either run two ranks or fail loudly -- never print an N=1 line (round-1 defect).
    No GPU here: the self-launcher refuses; a launcher/flag mismatch refuses too."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout) and '"metric"' not in r.stdout


def _diameter_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import oracle
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.RegressionNetwork.geomloss.samples_loss import global_range
    init_distributed()
    x, y = _diameter_case()
    h = x.shape[0] // world
    xs, ys = x[rank * h:(rank + 1) * h], y[rank * h:(rank + 1) * h]
    r = global_range(xs, ys)                      # the product's collective: one 2-float all-reduce, no host sync
    M = oracle.anchor_cost_matrix(x.shape[1])
    synced, aux = oracle.samples_loss(xs, ys, M, blur=.05, diameter=float(r[1] - r[0]), return_aux=True)
    # what a rank that only scans its own shard would derive
    _, aux_l = oracle.samples_loss(xs, ys, M, blur=.05, return_aux=True)
    np.savez(os.path.join(out_dir, "d%d.npz" % rank), range=r.numpy(), synced=synced.numpy(),
             eps_synced=np.asarray(aux["eps_s"]), eps_local=np.asarray(aux_l["eps_s"]))
    dist.destroy_process_group()


def _diameter_case():
    g = torch.Generator().manual_seed(5)
    B, N = 4, 32
    x = torch.softmax(torch.randn(B, N, generator=g), 1).view(B, N, 1)
    y = torch.softmax(3 * torch.randn(B, N, generator=g), 1).view(B, N, 1)
    y[B // 2:] = torch.softmax(8 * torch.randn(B // 2, N, generator=g), 1).view(B // 2, N, 1)   # rank 1's shard has the peaks
    return x, y


@pytest.mark.timeout(600)
def test_sync_diameter_two_ranks_reproduce_the_single_process_loss(tmp_path):
    """sinkhorn_divergence.py:9-18 takes the diameter over the WHOLE batch.  Two ranks x B/2 with the 2-float min/max
    all-reduce (``global_range``) reproduce the single-process eps-schedule bit for bit and the loss of the
    B samples; a rank that scans only its shard derives a different schedule for the shard with the smaller range."""
    import oracle
    port = _free_port()
    mp.spawn(_diameter_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x, y = _diameter_case()
    want, aux = oracle.samples_loss(x, y, oracle.anchor_cost_matrix(x.shape[1]), blur=.05, return_aux=True)
    want, eps_want = want.numpy(), np.asarray(aux["eps_s"])
    lo, hi = float(min(x.min(), y.min())), float(max(x.max(), y.max()))
    d = [np.load(tmp_path / ("d%d.npz" % k)) for k in range(2)]
    for k in range(2):
        np.testing.assert_array_equal(d[k]["range"], np.array([lo, hi], np.float32))
        np.testing.assert_array_equal(d[k]["eps_synced"], eps_want)       # the single-process eps-schedule, bit for bit
    np.testing.assert_allclose(np.concatenate([d[0]["synced"], d[1]["synced"]]), want, rtol=0, atol=1e-9)
    # rank 0's shard has the smaller range: scanning it alone gives another schedule (the converged loss barely moves --
    # the last eps is blur^p either way -- but the annealing path, and with it every intermediate dual, is not the reference's)
    assert len(d[0]["eps_local"]) != len(eps_want) or not np.array_equal(d[0]["eps_local"], eps_want)
