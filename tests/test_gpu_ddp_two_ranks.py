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


def _spade_case(seed=7, B=4, C=16, H=16, W=32):
    """Inputs + a SPADE module (parameter-free sync BatchNorm, normalization.py:68-115) with seeded weights."""
    from emlight_amd.GenProjector import networks
    torch.manual_seed(seed)
    mod = networks.SPADE("spadesyncbatch3x3", C, 3)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g) * 2.0 + 0.5
    x[B // 2:] += 1.5          # the two halves have different statistics: a per-rank norm would be visibly wrong
    seg = torch.rand(B, 3, 128, 256, generator=g)
    wy = torch.randn(B, C, H, W, generator=g)
    return mod, x, seg, wy


def _spade_run(mod, x, seg, wy):
    if x.shape[1] % 64 == 0:   # wide enough for the one-launch path (gamma|beta SphereConv + modulation epilogue): force it
        from emlight_amd.GenProjector.spherenet import SphereConv2D
        SphereConv2D.fused_min_bytes = 0
    mod = mod.cuda().train()
    x = x.cuda().requires_grad_(True)
    y = mod(x, seg.cuda(), slope=0.2)
    (y * wy.cuda()).sum().backward()
    bn = mod.param_free_norm
    out = {"y": y.detach(), "dx": x.grad, "running_mean": bn.running_mean, "running_var": bn.running_var}
    out.update({"grad/" + k: p.grad for k, p in mod.named_parameters()})
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def _spade_worker(rank, world, port, out_dir, C=16):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world), EML_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from emlight_amd.RegressionNetwork.engine import init_distributed
    init_distributed()
    mod, x, seg, wy = _spade_case(C=C)
    h = x.shape[0] // world
    sl = slice(rank * h, (rank + 1) * h)
    np.savez(os.path.join(out_dir, "spade%d.npz" % rank), **_spade_run(mod, x[sl], seg[sl], wy[sl]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("C", [16, 64])
def test_sync_bn_two_ranks_equal_one_rank_with_the_whole_batch(tmp_path, C, monkeypatch):
    """What sync_batchnorm/batchnorm.py:105-145 guarantees: 2 ranks x B/2 samples normalise with the statistics of the
    WHOLE batch.  SPADE's modulation (the HIP kernels + the (2C+1)-float all-reduce of its sums, forward and backward)
    on two ranks against ONE process holding both halves: outputs, input gradients, the summed parameter gradients and the
    running statistics agree to f32 round-off (the sums are f64; a per-rank norm would be off by O(1) here).  C = 64: through
    the one-launch SPADE (the gamma|beta SphereConv with the modulation as its epilogue), whose backward all-reduces the same sums."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    from emlight_amd.GenProjector.spherenet import SphereConv2D
    monkeypatch.setattr(SphereConv2D, "fused_min_bytes", SphereConv2D.fused_min_bytes)   # _spade_run may force the fused kernels
    mp.spawn(_spade_worker, args=(2, port, str(tmp_path), C), nprocs=2, join=True)
    want = _spade_run(*_spade_case(C=C))
    r = [np.load(tmp_path / ("spade%d.npz" % k)) for k in range(2)]
    for k in ("y", "dx"):
        got = np.concatenate([r[0][k], r[1][k]], 0)
        np.testing.assert_allclose(got, want[k], rtol=2e-5, atol=2e-5 * np.abs(want[k]).max(), err_msg=k)
    for k in ("running_mean", "running_var"):
        np.testing.assert_allclose(r[0][k], want[k], rtol=1e-6, atol=1e-7, err_msg=k)
        np.testing.assert_array_equal(r[0][k], r[1][k])
    for k in [k for k in want if k.startswith("grad/")]:
        got = r[0][k] + r[1][k]      # DDP would average; the sum of the per-rank gradients is the whole batch's
        np.testing.assert_allclose(got, want[k], rtol=1e-4, atol=2e-5 * np.abs(want[k]).max(), err_msg=k)
    # and the statistics really are the whole batch's, not a rank's own
    mod, x, seg, wy = _spade_case(C=C)
    half = _spade_run(mod, x[:2], seg[:2], wy[:2])
    assert np.abs(half["y"] - want["y"][:2]).max() > 1e-2


def _diameter_case():
    g = torch.Generator().manual_seed(5)
    B, N = 8, 128
    x = torch.softmax(torch.randn(B, N, generator=g), 1).view(B, N, 1)
    y = torch.softmax(3 * torch.randn(B, N, generator=g), 1).view(B, N, 1)
    y[B // 2:] = torch.softmax(8 * torch.randn(B // 2, N, generator=g), 1).view(B // 2, N, 1)   # rank 1's shard has the peaks
    return x, y


def _diameter_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world), EML_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    init_distributed()
    x, y = _diameter_case()
    h = x.shape[0] // world
    xs, ys = x[rank * h:(rank + 1) * h].cuda(), y[rank * h:(rank + 1) * h].cuda()
    out = {}
    for name, sync in (("synced", True), ("local", False)):
        r = SamplesLoss("sinkhorn", p=2, blur=.05, anchors=x.shape[1], sync_diameter=sync).forward_raw(xs, ys)
        out[name + "_loss"] = r["loss"].cpu().numpy()
        out[name + "_eps"] = r["eps_s"][:int(r["n_eps"].item())].cpu().numpy()
        out[name + "_gx"] = r["gx"].cpu().numpy()
    np.savez(os.path.join(out_dir, "diam%d.npz" % rank), **out)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sync_diameter_two_ranks_equal_one_rank_with_the_whole_batch(tmp_path):
    """The HIP Sinkhorn under data parallelism: with ``sync_diameter`` two ranks x B/2 derive the eps-schedule of the whole
    batch (bit for bit) and reproduce the one-process loss / gradient of the B samples (sinkhorn_divergence.py:9-18);
    without it rank 0, whose shard has the smaller range, runs another schedule."""
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_diameter_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x, y = _diameter_case()
    w = SamplesLoss("sinkhorn", p=2, blur=.05, anchors=x.shape[1]).forward_raw(x.cuda(), y.cuda())
    eps = w["eps_s"][:int(w["n_eps"].item())].cpu().numpy()
    d = [np.load(tmp_path / ("diam%d.npz" % k)) for k in range(2)]
    for k in range(2):
        np.testing.assert_array_equal(d[k]["synced_eps"], eps)
    np.testing.assert_allclose(np.concatenate([d[0]["synced_loss"], d[1]["synced_loss"]]), w["loss"].cpu().numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.concatenate([d[0]["synced_gx"], d[1]["synced_gx"]]), w["gx"].cpu().numpy(), rtol=1e-6, atol=1e-10)
    assert len(d[0]["local_eps"]) != len(eps) or not np.array_equal(d[0]["local_eps"], eps)


@pytest.mark.timeout(900)
def test_bench_two_ranks_runs_every_leg_without_deadlock():
    """`bench.py --gpus 2` end to end on the test box's one GPU (gloo transport, EML_SHARE_GPUS=1), small batches: every leg
    of the line -- the regression step with its live kernel-family timing, the projector leg with its instrumented step
    (which contains DDP's and SPADE's all-reduces: every rank has to run it), the joint leg -- completes on both ranks and
    rank 0 prints ONE N = 2 line.  (Round 3 found and fixed a rank-0-only instrumented step that would have hung N > 1.)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, EML_DIST_BACKEND="gloo", EML_SHARE_GPUS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--projector_batch", "2", "--joint_batch", "2", "--no_cpu_baseline"],
                       env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 8
    assert j["value"] > 0 and j["projector"]["value"] > 0 and j["joint"]["value"] > 0
    assert j["projector"]["config"]["global_batch"] == 4 and "kernel_families" in j["projector"] and "roofline" in j


def _count_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world), EML_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.GenProjector import networks
    from emlight_amd.GenProjector.data import projector_batch
    from emlight_amd.GenProjector.model_trainer import Trainer
    r, local, w = init_distributed()
    torch.manual_seed(0)
    tr = Trainer(networks.default_options(ngf=4, ndf=4), device="cuda:0", world=w)
    data = projector_batch(1, "cuda:0", seed=50 + rank)
    log = []          # (phase, dtype, numel) of every Python-level all-reduce = SPADE's sync-BN sums
    buckets = {"G": 0, "D": 0}
    real = dist.all_reduce

    def counting(t, *a, **k):
        if not k.get("async_op"):          # (the asynchronous ones are GradientBuckets' gradient buckets, counted below)
            log.append((phase[0], str(t.dtype), t.numel()))
        return real(t, *a, **k)

    def hook(name):
        def h(state, bucket):
            buckets[name] += 1
            return default_hooks.allreduce_hook(state, bucket)
        return h

    def counted_launch(name, red):
        inner = red._launch

        def launch(b):
            buckets[name] += 1
            return inner(b)
        red._launch = launch
    if tr.bucketsG is not None:            # the package's reducer (default) ...
        counted_launch("G", tr.bucketsG)
        counted_launch("D", tr.bucketsD)
    else:                                  # ... or torch's DistributedDataParallel (EML_DP_BUCKETS=0)
        tr._ddpG.register_comm_hook(None, hook("G"))
        tr._ddpD.register_comm_hook(None, hook("D"))
    phase = ["warm"]
    tr.step(data)                      # builds DDP's buckets (the first iteration may rebuild them)
    dist.all_reduce = counting
    try:
        buckets.update(G=0, D=0)
        phase[0] = "g_step"
        tr.run_generator_one_step(data)
        g_buckets = dict(buckets)
        buckets.update(G=0, D=0)
        phase[0] = "d_step"
        tr.run_discriminator_one_step(data)
        d_buckets = dict(buckets)
    finally:
        dist.all_reduce = real
    # SPADE's sums: f64 vectors of 2C+1 entries; anything else that reaches the Python-level all_reduce (DDP's own
    # bookkeeping) is reported, not counted
    spade = [(p, n) for p, dt, n in log if dt == "torch.float64" and n % 2 == 1 and n >= 9]
    other = len(log) - len(spade)
    n_g = sum(1 for p, n in spade if p == "g_step")
    n_d = sum(1 for p, n in spade if p == "d_step")
    np.save(os.path.join(out_dir, "c%d.npy" % rank),
            np.array([n_g, n_d, g_buckets["G"], g_buckets["D"], d_buckets["G"], d_buckets["D"], other, max(n for p, n in spade)]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_collective_counts_of_a_projector_iteration(tmp_path):
    """what one projector iteration puts on the wire besides DDP's gradient buckets.  SPADE's parameter-free
    BatchNorm is SYNCHRONISED (sync_batchnorm/batchnorm.py:105-126): per generator pass the forward all-reduces one
    (2C+1)-f64 vector per distinct normalised input -- norm_0 / norm_s of a block share theirs: 2 per block, 14 in all --
    and the backward one per norm (18: every normalised input carries a gradient, the head block's through the crop
    encoder).  The generator step holds D out of the graph (no D buckets); the discriminator step runs G without grad
    (forward statistics only, no G buckets).  Counted on the wire here, not assumed."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_count_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "c0.npy"), np.load(tmp_path / "c1.npy")
    np.testing.assert_array_equal(c0, c1)
    n_g, n_d, gG, gD, dG, dD, other, widest = c0
    blocks, shortcuts = 7, 4                       # SPADEGenerator: head_0, G_middle_0/1, up_0..3; up_* have a learned shortcut
    fwd = 2 * blocks                               # norm_0 (+ norm_s on the same sums) and norm_1
    assert n_d == fwd, "discriminator step: the no-grad generator pass all-reduces its %d forward statistics only (got %d)" % (fwd, n_d)
    assert n_g == fwd + 2 * blocks + shortcuts, "generator step: %d forward + %d backward sync-BN all-reduces expected, got %d" % (
        fwd, 2 * blocks + shortcuts, n_g)
    assert widest == 2 * 64 + 1 and other <= 4       # (2C+1) f64 sums, C <= 16 * ngf = 64 here; DDP's own small reductions
    assert gG >= 1 and gD == 0 and dG == 0 and dD >= 1


def _rccl_single_worker(rank, dry_run, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    os.environ.pop("EML_DIST_BACKEND", None)           # the default backend on a GPU: "nccl" = RCCL
    if dry_run:
        os.environ["EML_DIST_SINGLE"] = "1"
    else:
        os.environ.pop("EML_DIST_SINGLE", None)
    import torch.distributed as dist
    from emlight_amd import _dist
    from emlight_amd.GenProjector import networks
    from emlight_amd.RegressionNetwork.engine import init_distributed
    from emlight_amd.joint import JointTrainer, joint_batch
    # the generator's crop encoder is nn.Conv2d -> MIOpen, whose default weight-gradient algorithm sums with atomics: two PLAIN
    # runs already differ by up to 5e-6 on GAN_Feat (tools/exp/ddp_flake_probe.py); with MIOpen's deterministic algorithms
    # every run, plain or data-parallel, gives the same bits (the package's own kernels have no atomics)
    torch.backends.cudnn.deterministic = True
    r, local, w = init_distributed()
    assert dist.is_initialized() == bool(dry_run) and _dist.dp_active() == bool(dry_run)
    if dry_run:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    calls = {"n": 0}
    orig = dist.all_reduce

    def counting(t, *a, **k):
        calls["n"] += 1
        return orig(t, *a, **k)
    dist.all_reduce = counting
    torch.manual_seed(0)
    tr = JointTrainer(networks.default_options(ngf=4, ndf=4), anchors=32, crop_hw=(64, 96), device="cuda:0", world=w)
    assert (tr.reg.buckets is not None or tr.reg.ddp is not None) == bool(dry_run)
    assert (tr.proj.bucketsG is not None or hasattr(tr.proj, "_ddpG")) == bool(dry_run)
    batch = joint_batch(2, "cuda:0", 32, (64, 96), seed=9)
    for _ in range(2):
        losses = tr.step(batch)
    flat = torch.cat([q.detach().reshape(-1) for net in (tr.reg.model, tr.proj.model.netG, tr.proj.model.netD) for q in net.parameters()])
    vals = np.array([float(v.mean()) for v in losses.values()] + [float(flat.double().sum()), float(flat.abs().max()), calls["n"]])
    np.save(os.path.join(out_dir, "s%d.npy" % int(dry_run)), vals)
    if dist.is_initialized():
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_one_rank_rccl_dry_run_of_the_data_parallel_path(tmp_path):
    """``EML_DIST_SINGLE=1`` (emlight_amd/_dist.py): the whole data-parallel machinery of a joint
    iteration -- process group on RCCL ("nccl"), DistributedDataParallel around encoder / generator / discriminator, SPADE's
    synchronised BatchNorm all-reduces, the Sinkhorn diameter's -- with ONE rank on the one GPU of the test box (RCCL refuses two
    ranks per device: "Duplicate GPU detected", profiles/r06_rccl_probe.txt).  A one-rank all-reduce is the identity, so two
    iterations must give exactly the losses and parameters of the plain single-process run; and the collectives really ran."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    for dry in (0, 1):
        mp.spawn(_rccl_single_worker, args=(dry, port + dry, str(tmp_path)), nprocs=1, join=True)
    plain, dry = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert plain[-1] == 0 and dry[-1] >= 2 * (14 + 18 + 14), (plain[-1], dry[-1])   # sync-BN sums of two iterations (+ diameters)
    assert np.isfinite(dry).all()
    # DDP hands the optimizer gradients from its buckets: the same numbers (a one-rank mean), summed in the same kernels
    np.testing.assert_allclose(dry[:-1], plain[:-1], rtol=1e-6, atol=1e-7)   # measured: identical bits, 4 + 4 fresh processes
