"""GPU parity: HIP DenseNet-BC engine (through the C ABI) vs the oracle / the reference's golden vectors.

north_star gate: regressed (distribution, intensity, RGB, ambient) tensors within 1e-4 abs."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
OUT_ATOL = 1e-4
# Sampled-entry error of the golden train-step gradients in units of the tensor's RMS gradient.  Measured on the
# MI355X: worst 6.8e-3 (conv0.weight), then 5.9e-3 (denseblock1.denselayer1.conv1.weight), 5.0e-3 (transition1.conv.weight);
# head tensors 3e-6.  (With BatchNorm sums pre-accumulated over 4 pixels in f32 the worst was 3.3e-2 -- ADVICE round 1 was
# right; they accumulate per element in f64 again.)  What remains is the f32 conditioning of this network's gradient:
# test_gradient_error_is_f32_conditioning below shows the reference's own stock-op graph in f32 is just as far from the
# f64 gradient (relative L2 per tensor: median 4e-4..3e-3, up to 1e-2..3e-2 on the norm weights).
GOLDEN_GRAD_MAX, GOLDEN_GRAD_MEDIAN = 2e-2, 8e-3
GRAD_L2_MAX, GRAD_L2_MEDIAN = 2e-2, 4e-3   # per-tensor relative L2 vs the f32 oracle at toy sizes; measured max 9.6e-3, median 2.4e-3
KEYS = ("distribution", "intensity", "rgb_ratio", "ambient")


def _pair(anchors, crop_hw, seed):
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    ref = oracle.OracleDenseNet(anchors=anchors, crop_hw=crop_hw)
    sd = oracle.deterministic_state_dict(ref.state_dict(), seed=seed)
    ref.load_state_dict(sd)
    net = DenseNet(anchors=anchors, crop_hw=crop_hw).cuda()
    net.load_state_dict(sd)
    return ref, net


@pytest.mark.parametrize("crop_hw,B", [((64, 96), 2), ((32, 32), 3), ((96, 160), 1)])
@pytest.mark.parametrize("train", [False, True])
def test_forward_small_vs_oracle(crop_hw, B, train):
    ref, net = _pair(32, crop_hw, seed=3)
    ref.train(train), net.train(train)
    x = torch.from_numpy(np.random.default_rng([1, B]).random((B, 3) + crop_hw, dtype=np.float32))
    with torch.no_grad():
        want, got = ref(x), net(x.cuda())
        fw = torch.nn.functional.avg_pool2d(torch.relu(ref.features_forward(x)), 4).reshape(B, -1)
        fg = net.pooled_features(x.cuda())
    np.testing.assert_allclose(fg.cpu().numpy(), fw.numpy(), rtol=3e-5, atol=2e-5)
    for k in KEYS:
        # synthetic eval-mode weights give outputs of O(100): 1e-4 abs + 1e-5 rel
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), rtol=1e-5, atol=OUT_ATOL)
    if train:  # running statistics follow nn.BatchNorm2d
        for name in ("features.norm0", "features.denseblock2.denselayer5.norm2", "features.last_norm3",
                     "features.transition1.norm"):
            a, b = dict(ref.named_modules())[name], dict(net.named_modules())[name]
            np.testing.assert_allclose(b.running_mean.cpu().numpy(), a.running_mean.numpy(), rtol=0, atol=1e-5)
            np.testing.assert_allclose(b.running_var.cpu().numpy(), a.running_var.numpy(), rtol=1e-4, atol=1e-6)
            assert int(b.num_batches_tracked) == 2  # two train-mode forward calls above (nn.BatchNorm2d counts calls)


def test_forward_golden_reference_geometry(golden_densenet):
    """192x256 crops, 96 anchors: outputs of the REAL reference DenseNet (eval and train mode)."""
    g = golden_densenet
    _, net = _pair(96, (192, 256), seed=0)
    x = torch.from_numpy(np.random.default_rng([0]).random((2, 3, 192, 256), dtype=np.float32)).cuda()
    with torch.no_grad():
        net.eval()
        pe = net(x)
        net.train()
        pt = net(x)
    for k in KEYS:
        np.testing.assert_allclose(pe[k].cpu().numpy(), g["eval/" + k], rtol=0, atol=OUT_ATOL)
        np.testing.assert_allclose(pt[k].cpu().numpy(), g["train/" + k], rtol=0, atol=OUT_ATOL)


def test_forward_cfg1_240x320_128_anchors(golden_densenet):
    g = golden_densenet
    _, net = _pair(128, (240, 320), seed=1)
    net.eval()
    x = torch.from_numpy(np.random.default_rng([2]).random((1, 3, 240, 320), dtype=np.float32)).cuda()
    with torch.no_grad():
        p = net(x)
    for k in KEYS:
        np.testing.assert_allclose(p[k].cpu().numpy(), g["cfg1/" + k], rtol=0, atol=OUT_ATOL)


def test_golden_train_step_reference_geometry(golden_densenet):
    """One full training step of the REAL reference (192x256 crops, 96 anchors, B=2, seed-0 weights, blur .025;
    tests/golden/make_golden.py::gen_densenet) through the product: HIP encoder forward, HIP Sinkhorn, the product's
    ``regression_loss``, HIP backward, Adam -- loss terms, the 12 sampled parameter gradients, running statistics and
    post-step weights.  Exactly what tests/test_oracle_golden.py holds the CPU oracle to."""
    from emlight_amd.RegressionNetwork.engine import RegressionTrainer
    g = golden_densenet
    tr = RegressionTrainer(anchors=96, crop_hw=(192, 256), blur=.025, device="cuda:0")
    ref = oracle.OracleDenseNet()
    tr.model.load_state_dict(oracle.deterministic_state_dict(ref.state_dict(), seed=0))
    x = torch.from_numpy(np.random.default_rng([0]).random((2, 3, 192, 256), dtype=np.float32)).cuda()
    batch = {k: torch.from_numpy(g["train/gt_" + k]).cuda() for k in KEYS}
    batch["crop"] = x
    loss, terms = tr.step(batch)
    for k in KEYS:
        np.testing.assert_allclose(tr.last_pred[k].detach().cpu().numpy(), g["train/" + k], rtol=0, atol=OUT_ATOL)
    np.testing.assert_allclose(np.array([float(t.detach()) for t in terms.values()]), g["train/loss_terms"], rtol=1e-4)
    named = dict(tr.model.named_parameters())
    worst = []
    for key in [k[len("train/grad/"):] for k in g.z.files if k.startswith("train/grad/")]:
        got = named[key].grad.detach().reshape(-1).cpu()[torch.from_numpy(g["train/grad_idx/" + key])].numpy()
        want = g["train/grad/" + key]
        l2 = float(g["train/grad_l2/" + key])
        # error of the sampled entries in units of the tensor's RMS gradient (l2 / sqrt(numel))
        rms = l2 / np.sqrt(named[key].numel())
        worst.append((float(np.abs(got - want).max() / rms), key))
    worst.sort(reverse=True)
    print("golden train step: worst sampled |dgrad| / rms(grad) per tensor:", worst)
    # two f32 evaluations of a 100-BN-layer train-mode network agree to ~1e-2 of the RMS gradient per entry (see the
    # constants above and test_gradient_error_is_f32_conditioning); the bound is on sampled entries
    assert worst[0][0] < GOLDEN_GRAD_MAX, worst[:4]
    assert np.median([e for e, _ in worst]) < GOLDEN_GRAD_MEDIAN, worst
    np.testing.assert_allclose(tr.model.features.norm0.running_mean.cpu().numpy(), g["train/running_mean/features.norm0"],
                               atol=1e-6)
    np.testing.assert_allclose(tr.model.features.last_norm3.running_var.cpu().numpy(),
                               g["train/running_var/features.last_norm3"], rtol=1e-4)
    # Adam's first step moves every entry by lr * g / (|g| + 1e-8): +-1e-4 unless the gradient is ~1e-8
    np.testing.assert_allclose(tr.model.fc_dist.bias.detach().cpu().numpy(), g["train/post_step/fc_dist.bias"], rtol=0,
                               atol=2e-6)


def test_golden_train_step_baseline_geometry(golden_densenet_cfg2):
    """The backward AT BASELINE's geometry against the reference: one full training step of the reference class at
    240x320 crops / 128 anchors / blur .05 (B = 2; fc and fc_dist swapped for matching nn.Linear, SURVEY 8c;
    tests/golden/make_golden.py::gen_densenet_cfg2) through RegressionTrainer.step -- block 1 runs at 240x320, where the
    192x256 golden step never exercised the kernels' tile boundaries (320 = 10 x 32, 240 = 30 x 8 row tiles; pixel count
    153 600 per image).  Same quantities and bounds as the reference-geometry step above; 17 sampled gradient tensors."""
    from emlight_amd.RegressionNetwork.engine import RegressionTrainer
    g = golden_densenet_cfg2
    tr = RegressionTrainer(anchors=128, crop_hw=(240, 320), blur=.05, device="cuda:0")
    ref = oracle.OracleDenseNet(anchors=128, crop_hw=(240, 320))
    tr.model.load_state_dict(oracle.deterministic_state_dict(ref.state_dict(), seed=5))
    x = torch.from_numpy(np.random.default_rng([40]).random((2, 3, 240, 320), dtype=np.float32)).cuda()
    batch = {k: torch.from_numpy(g["train/gt_" + k]).cuda() for k in KEYS}
    batch["crop"] = x
    loss, terms = tr.step(batch)
    for k in KEYS:
        np.testing.assert_allclose(tr.last_pred[k].detach().cpu().numpy(), g["train/" + k], rtol=0, atol=OUT_ATOL)
    np.testing.assert_allclose(np.array([float(t.detach()) for t in terms.values()]), g["train/loss_terms"], rtol=1e-4)
    named = dict(tr.model.named_parameters())
    worst = []
    for key in [k[len("train/grad/"):] for k in g.z.files if k.startswith("train/grad/")]:
        got = named[key].grad.detach().reshape(-1).cpu()[torch.from_numpy(g["train/grad_idx/" + key])].numpy()
        rms = float(g["train/grad_l2/" + key]) / np.sqrt(named[key].numel())
        worst.append((float(np.abs(got - g["train/grad/" + key]).max() / rms), key))
    worst.sort(reverse=True)
    print("golden train step at 240x320: worst sampled |dgrad| / rms(grad) per tensor:", worst)
    assert worst[0][0] < GOLDEN_GRAD_MAX, worst[:4]
    assert np.median([e for e, _ in worst]) < GOLDEN_GRAD_MEDIAN, worst
    f = tr.model.features
    np.testing.assert_allclose(f.norm0.running_mean.cpu().numpy(), g["train/running_mean/features.norm0"], atol=1e-6)
    np.testing.assert_allclose(f.denseblock2.denselayer3.norm2.running_var.cpu().numpy(),
                               g["train/running_var/features.denseblock2.denselayer3.norm2"], rtol=1e-4)
    np.testing.assert_allclose(f.last_norm3.running_var.cpu().numpy(), g["train/running_var/features.last_norm3"], rtol=1e-4)
    np.testing.assert_allclose(tr.model.fc_dist.bias.detach().cpu().numpy(), g["train/post_step/fc_dist.bias"], rtol=0,
                               atol=2e-6)


@pytest.mark.parametrize("crop_hw,B,anchors", [((192, 256), 2, 96), ((64, 96), 2, 32)])
def test_gradient_error_is_f32_conditioning(crop_hw, B, anchors):
    """Whole-network gradients of the HIP engine (f32) and of the oracle's stock-op graph in f32 are both compared with
    the oracle in f64 on the same weights and input.  The HIP engine must be as close to the f64 gradient as the
    reference's own f32 arithmetic is (per-tensor relative L2: median and worst tensor within 3x of the stock-op f32
    graph's) -- the residual 1e-3..1e-2 is conditioning, shared by any f32 implementation.  The analytically
    zero gradients (last_norm{1,2}.bias feed only train-mode BNs) are excluded."""
    ref32, net = _pair(anchors, crop_hw, seed=0)
    ref64 = oracle.OracleDenseNet(anchors=anchors, crop_hw=crop_hw).double()
    ref64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in ref32.state_dict().items()})
    ref32, ref64 = ref32.cuda().train(), ref64.cuda().train()   # stock ops on the GPU (fast f64); same maths as on CPU
    net.train()
    g = np.random.default_rng(0)
    x = torch.from_numpy(g.random((B, 3) + crop_hw, dtype=np.float32)).cuda()
    w = {k: torch.from_numpy(g.standard_normal(s).astype(np.float32)).cuda()
         for k, s in (("distribution", (B, anchors)), ("intensity", (B, 1)), ("rgb_ratio", (B, 3)), ("ambient", (B, 3)))}
    for m, xx in ((ref32, x), (ref64, x.double()), (net, x)):
        out = m(xx)
        sum((out[k] * w[k].to(out[k].dtype)).sum() for k in KEYS).backward()
    rms = lambda a: float(np.sqrt(np.mean(np.square(a))))
    n64, n32 = dict(ref64.named_parameters()), dict(ref32.named_parameters())
    e_hip, e_o32 = [], []
    for name, p in net.named_parameters():
        if name in ("features.last_norm1.bias", "features.last_norm2.bias"):
            continue
        t = n64[name].grad.cpu().numpy()
        e_hip.append(rms(p.grad.cpu().numpy().astype(np.float64) - t) / rms(t))
        e_o32.append(rms(n32[name].grad.cpu().numpy().astype(np.float64) - t) / rms(t))
    print("rel-L2 vs f64: HIP median %.2e max %.2e | stock f32 median %.2e max %.2e"
          % (np.median(e_hip), max(e_hip), np.median(e_o32), max(e_o32)))
    # run-to-run the stock-op graph itself moves by tens of per cent in these figures (its library kernels reorder sums)
    assert np.median(e_hip) <= 3.0 * np.median(e_o32) + 5e-4
    assert max(e_hip) <= 3.0 * max(e_o32) + 5e-4


def test_two_forwards_before_backward_keep_their_own_activations():
    """loss(model(a)) + loss(model(b)) with ONE backward: the second forward must not overwrite the activations the
    first graph still needs (each live graph owns its workspace)."""
    ref, net = _pair(32, (64, 96), seed=7)
    ref.train(), net.train()
    g = np.random.default_rng(11)
    xa = torch.from_numpy(g.random((2, 3, 64, 96), dtype=np.float32))
    xb = torch.from_numpy(g.random((2, 3, 64, 96), dtype=np.float32))
    (sum(v.sum() for v in ref(xa).values()) + 2.0 * sum(v.square().sum() for v in ref(xb).values())).backward()
    pa, pb = net(xa.cuda()), net(xb.cuda())
    (sum(v.sum() for v in pa.values()) + 2.0 * sum(v.square().sum() for v in pb.values())).backward()
    _grad_check(net, ref)
    assert len(net._hip._ws[next(iter(net._hip._ws))]) == 2   # two workspaces were alive at once
    # the next single forward/backward reuses a released workspace instead of allocating a third
    net.zero_grad(set_to_none=True)
    sum(v.sum() for v in net(xa.cuda()).values()).backward()
    assert len(net._hip._ws[next(iter(net._hip._ws))]) == 2


def _grad_check(net, ref):
    """Whole-network gradient parity.  Two f32 implementations of a ReLU network cannot agree
    element-wise: forward values differ by ~1e-5 rel, so a few pre-activations per layer with
    |pre| < 1e-5 get the opposite ReLU mask (measured: exactly one channel of transition3.norm.bias
    off by one element's gradient, every other channel < 5e-7) -- and, more importantly, the gradient of this network is
    only conditioned to ~1e-3 in f32 (test_gradient_error_is_f32_conditioning).  The bounds below are therefore on
    the relative L2 error per tensor, scaled by the network's typical gradient magnitude so that
    analytically-zero gradients (last_norm{1,2}.bias feed only train-mode BNs) do not blow up;
    the per-kernel tests in test_gpu_dense_kernels.py hold each kernel to f32 round-off."""
    named_r, named_g = dict(ref.named_parameters()), dict(net.named_parameters())
    rms = lambda a: float(np.sqrt(np.mean(np.square(a))))
    floor = 1e-3 * np.median([rms(p.grad.numpy()) for p in named_r.values()])
    errs = []
    for name, pr in named_r.items():
        gr, gg = pr.grad.numpy(), named_g[name].grad.cpu().numpy()
        assert np.isfinite(gg).all(), name
        errs.append((rms(gg - gr) / max(rms(gr), floor), name))
    errs.sort(reverse=True)
    print("whole-network gradient vs oracle: max rel-L2 %.2e (%s), median %.2e" % (errs[0][0], errs[0][1], np.median([e for e, _ in errs])))
    assert errs[0][0] < GRAD_L2_MAX, "largest relative L2 grad errors: %s" % errs[:8]
    assert np.median([e for e, _ in errs]) < GRAD_L2_MEDIAN, "median relative L2 grad error %g" % np.median([e for e, _ in errs])


@pytest.mark.parametrize("crop_hw,B", [((64, 96), 2), ((32, 64), 3)])
def test_backward_small_vs_oracle(crop_hw, B):
    """Every parameter gradient of one train-mode step vs the oracle's autograd (CPU)."""
    ref, net = _pair(32, crop_hw, seed=5)
    ref.train(), net.train()
    g = np.random.default_rng([4, B])
    x = torch.from_numpy(g.random((B, 3) + crop_hw, dtype=np.float32))
    w = {k: torch.from_numpy(g.standard_normal(s).astype(np.float32))
         for k, s in (("distribution", (B, 32)), ("intensity", (B, 1)), ("rgb_ratio", (B, 3)), ("ambient", (B, 3)))}
    po = ref(x)
    sum((po[k] * w[k]).sum() for k in KEYS).backward()
    pg = net(x.cuda())
    sum((pg[k] * w[k].cuda()).sum() for k in KEYS).backward()
    for k in KEYS:
        np.testing.assert_allclose(pg[k].detach().cpu().numpy(), po[k].detach().numpy(), rtol=1e-5, atol=OUT_ATOL)
    _grad_check(net, ref)


def test_conv0_kernel_choice_does_not_change_a_training_step(monkeypatch):
    """``EML_CONV0_MFMA=0 / 1`` (VALU kernel / matrix unit, the default): conv0's outputs and BatchNorm sums are bit-identical
    (tests/test_gpu_dense_kernels.py), so a whole train-mode step is -- every head and every parameter gradient, bit for bit."""
    from emlight_amd import _lib
    runs = []
    for knob in ("0", "1"):
        monkeypatch.setenv("EML_CONV0_MFMA", knob)
        seen = []
        for name in ("eml_dense_conv0_fwd_f32", "eml_dense_conv0_fwd_mfma_f32"):
            real = getattr(_lib.lib(), name)
            monkeypatch.setattr(_lib.lib(), name, (lambda r, n: lambda *a: (seen.append(n), r(*a))[1])(real, name))
        _, net = _pair(32, (64, 96), seed=5)
        net.train()
        g = np.random.default_rng([4, 2])
        x = torch.from_numpy(g.random((2, 3, 64, 96), dtype=np.float32)).cuda()
        out = net(x)
        sum(v.square().sum() for v in out.values()).backward()
        assert seen == ["eml_dense_conv0_fwd_mfma_f32" if knob == "1" else "eml_dense_conv0_fwd_f32"]
        runs.append(([out[k].detach().clone() for k in KEYS], [q.grad.clone() for q in net.parameters()]))
        monkeypatch.undo()
    for a, b in zip(runs[0][0] + runs[0][1], runs[1][0] + runs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("block_config,num_init", [((8, 5, 4), 24), ((2, 7, 4), 24), ((2, 2, 2), 24)])
def test_other_block_configs_vs_oracle(block_config, num_init):
    """The class signature's other depths (DenseNet.py:82-83: block_config) through the same schedule logic:
    pairs of layers with the compact top-24 hand-over where a next pair exists and the block's first channel is a multiple
    of 4, a single leftover layer for odd counts, blocks that start at channels which are not -- forward values and every
    parameter gradient of a train-mode step against the oracle's autograd (CPU)."""
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    crop_hw, B, anchors = (32, 64), 2, 32
    ref = oracle.OracleDenseNet(block_config=block_config, num_init_features=num_init, anchors=anchors, crop_hw=crop_hw)
    sd = oracle.deterministic_state_dict(ref.state_dict(), seed=11)
    ref.load_state_dict(sd)
    net = DenseNet(block_config=block_config, num_init_features=num_init, anchors=anchors, crop_hw=crop_hw).cuda()
    net.load_state_dict(sd)
    ref.train(), net.train()
    g = np.random.default_rng([9, len(block_config), num_init])
    x = torch.from_numpy(g.random((B, 3) + crop_hw, dtype=np.float32))
    w = {k: torch.from_numpy(g.standard_normal(s).astype(np.float32))
         for k, s in (("distribution", (B, anchors)), ("intensity", (B, 1)), ("rgb_ratio", (B, 3)), ("ambient", (B, 3)))}
    po = ref(x)
    sum((po[k] * w[k]).sum() for k in KEYS).backward()
    pg = net(x.cuda())
    sum((pg[k] * w[k].cuda()).sum() for k in KEYS).backward()
    for k in KEYS:
        np.testing.assert_allclose(pg[k].detach().cpu().numpy(), po[k].detach().numpy(), rtol=1e-5, atol=OUT_ATOL)
    _grad_check(net, ref)


def test_configurations_outside_the_envelope_are_refused_up_front():
    """What the kernels are not built for raises when the engine is made -- not in the middle of a backward pass."""
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    for kw in (dict(num_init_features=32), dict(num_init_features=16), dict(block_config=(4, 5, 2)),   # transition 2: 96 -> 48
               dict(growth_rate=16), dict(bn_size=2), dict(block_config=(32, 16, 16))):
        with pytest.raises(NotImplementedError):
            DenseNet(anchors=8, crop_hw=(32, 64), **kw)     # the constructor asks HipDenseEncoder.check_supported


def test_properties_at_full_baseline_size():
    """BASELINE configs[1] (B=64, 240x320, 128 anchors) is too large for the oracle, so the full-size check uses
    properties of the encoder that do not depend on size: bitwise run-to-run determinism, sample independence in eval
    mode, batch-permutation equivariance in train mode (batch statistics are order-free), and exact linearity of the
    backward in the upstream gradient."""
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    torch.manual_seed(5)
    net = DenseNet(anchors=128, crop_hw=(240, 320)).cuda()
    ref = oracle.OracleDenseNet(anchors=128, crop_hw=(240, 320))
    net.load_state_dict(oracle.deterministic_state_dict(ref.state_dict(), seed=2))
    x = torch.rand(64, 3, 240, 320, device="cuda")

    net.eval()
    with torch.no_grad():
        a, b = net(x), net(x)
        sub = net(x[8:12].contiguous())
    for k in KEYS:
        assert torch.equal(a[k], b[k]), "eval forward must be run-to-run exact (%s)" % k
        np.testing.assert_allclose(a[k][8:12].cpu().numpy(), sub[k].cpu().numpy(), rtol=1e-5, atol=1e-5,
                                   err_msg="eval outputs of a sample must not depend on its batch (%s)" % k)

    net.train()
    perm = torch.randperm(64, device="cuda")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    out = net(x)
    net.load_state_dict(state)  # same running statistics for the second pass
    outp = net(x[perm].contiguous())
    for k in KEYS:
        scale = float(out[k].detach().abs().max())
        np.testing.assert_allclose(outp[k].detach().cpu().numpy(), out[k].detach()[perm].cpu().numpy(), rtol=0,
                                   atol=2e-5 * max(scale, 1.0), err_msg="batch permutation (%s)" % k)

    def grads(factor):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        o = net(x)
        (factor * sum((o[k] * (i + 1)).sum() for i, k in enumerate(KEYS))).backward()
        return [p.grad.clone() for p in net.parameters()]
    g1, g2 = grads(1.0), grads(2.0)
    for p1, p2 in zip(g1, g2):
        assert torch.equal(2.0 * p1, p2), "backward must be exactly linear in the upstream gradient"
    assert all(bool(torch.isfinite(g).all()) for g in g1)


def test_cfg2_full_batch_values_against_the_stock_op_graph():
    """BASELINE configs[1] at its FULL size (B = 64, 240 x 320 crops, 128 anchors, train-mode BatchNorm) against the oracle's
    stock-op graph -- the reference's arithmetic (RegressionNetwork/train.py:81-102, DenseNet.py:135-157) -- run on the GPU in
    f32 and, as the yardstick for the gradients, in f64, on the same weights, input and targets (VERDICT round 5, item 7:
    the full-size test above checks properties, values were pinned at B = 2 only).  Bounds: the four regressed tensors 1e-4
    abs (north_star), the loss terms 1e-4 rel, every parameter's gradient as close to the f64 gradient as the stock f32 graph
    is (relative L2 within 3x, like test_gradient_error_is_f32_conditioning).  The oracle re-runs each dense layer in its
    backward (``checkpoint_layers``: same values) so that the f64 graph fits in HBM."""
    import gc
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    from emlight_amd.RegressionNetwork.engine import regression_loss as hip_loss
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    B, anchors, crop_hw = 64, 128, (240, 320)
    ref32, net = _pair(anchors, crop_hw, seed=4)
    sd = ref32.state_dict()
    batch = synthetic_batch(B, anchors, crop_hw, seed=77, device="cuda")
    M = oracle.anchor_cost_matrix(anchors)      # the oracle's Sinkhorn is a CPU restatement: the loss is taken on host copies

    def stock(dtype):
        m = oracle.OracleDenseNet(anchors=anchors, crop_hw=crop_hw).to(dtype)
        m.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()})
        m = m.cuda().train()
        m.checkpoint_layers = True
        gt = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in batch.items()}
        pred = m(gt["crop"])
        pred_h = {k: v.cpu() for k, v in pred.items()}          # differentiable copies: the gradient flows back to the GPU graph
        gt_h = {k: v.cpu() for k, v in gt.items() if k != "crop"}
        loss, terms = oracle.regression_loss(pred_h, gt_h, lambda a, b: oracle.samples_loss(a, b, M.to(dtype), blur=.05), anchors)
        loss.backward()
        out = ({k: pred[k].detach().double().cpu() for k in KEYS}, {k: float(v) for k, v in terms.items()},
               {n: q.grad.detach().double().cpu() for n, q in m.named_parameters()})
        del m, pred, pred_h, gt_h, loss, terms, gt
        gc.collect()
        torch.cuda.empty_cache()
        return out
    o32, t32, g32 = stock(torch.float32)
    o64, t64, g64 = stock(torch.float64)
    net.train()
    pred = net(batch["crop"])
    loss, terms = hip_loss(pred, batch, SamplesLoss("sinkhorn", p=2, blur=.05, anchors=anchors), anchors)
    loss.backward()
    for k in KEYS:   # north_star: regressed tensors within 1e-4 abs -- against the f64 values and against the stock f32 graph
        got = pred[k].detach().double().cpu()
        assert float((got - o64[k]).abs().max()) <= OUT_ATOL, (k, float((got - o64[k]).abs().max()))
        assert float((got - o32[k]).abs().max()) <= OUT_ATOL, (k, float((got - o32[k]).abs().max()))
    for k, v in terms.items():
        assert abs(float(v) - t64[k]) <= 1e-4 * abs(t64[k]) + 1e-9, (k, float(v), t64[k], t32[k])
    rms = lambda a: float(a.square().mean().sqrt())
    e_hip, e_o32 = [], []
    for name, q in net.named_parameters():
        if name in ("features.last_norm1.bias", "features.last_norm2.bias"):   # analytically zero (feed train-mode BNs only)
            continue
        t = g64[name]
        e_hip.append(rms(q.grad.detach().double().cpu() - t) / rms(t))
        e_o32.append(rms(g32[name] - t) / rms(t))
    print("B=64 240x320: gradient rel-L2 vs f64: HIP median %.2e max %.2e | stock f32 median %.2e max %.2e"
          % (np.median(e_hip), max(e_hip), np.median(e_o32), max(e_o32)))
    assert np.median(e_hip) <= 3.0 * np.median(e_o32) + 5e-4
    assert max(e_hip) <= 3.0 * max(e_o32) + 5e-4


def test_training_curve_tracks_stock_op_oracle():
    """24 Adam steps (10x the reference learning rate) with the HIP engine and with the oracle's stock-op encoder (run
    on the same GPU) from the same initial weights and batches: the loss curves stay together.  (Element-wise agreement is impossible across two f32
    implementations of a ReLU network -- DESIGN.md section 4 -- so this bounds the drift of the whole training loop: measured <= 6 % per point.)"""
    from emlight_amd.RegressionNetwork.data import synthetic_batch
    from emlight_amd.RegressionNetwork.engine import RegressionTrainer
    batches = [synthetic_batch(4, 32, (64, 96), seed=100 + i, device="cuda:0") for i in range(4)]
    curves = {}
    sd = oracle.deterministic_state_dict(oracle.OracleDenseNet(anchors=32, crop_hw=(64, 96)).state_dict(), seed=9)
    for eng in ("aten", "hip"):
        model = oracle.OracleDenseNet(anchors=32, crop_hw=(64, 96)) if eng == "aten" else None
        tr = RegressionTrainer(anchors=32, crop_hw=(64, 96), blur=.05, device="cuda:0", lr=1e-3, model=model)
        tr.model.load_state_dict(sd)
        curves[eng] = np.array([float(tr.step(batches[i % 4])[0].detach()) for i in range(24)])
    dev = np.abs(curves["hip"] / curves["aten"] - 1.0)
    print("training curve: max relative deviation first 6 steps %.2e, all 24 steps %.2e" % (dev[:6].max(), dev.max()))
    assert curves["hip"][0] == pytest.approx(curves["aten"][0], rel=1e-5)
    np.testing.assert_allclose(curves["hip"][:6], curves["aten"][:6], rtol=0.01)
    # later steps: training at this rate is chaotic in f32 (a change in summation order moves single points by
    # several per cent), so the bound is loose; what matters is that the curves do not separate
    np.testing.assert_allclose(curves["hip"], curves["aten"], rtol=0.1)   # measured: 2.2e-3 over the first 6 steps, 5.8e-2 over all 24
    assert curves["hip"][-1] < 0.8 * curves["hip"][:8].max()  # and it trains


def test_dgamma_of_small_and_negative_gammas_is_recomputed_directly(monkeypatch):
    """BN1's dgamma comes from the conv's weight gradient, S2 = (sum_o W*dW - beta*S1) / gamma -- a quotient
    that amplifies the f32 error of its two terms by 1/|gamma|.  Trained DenseNets have |gamma| of 1e-6..1e-3 in some
    channels (every other test draws gamma from U(0.5, 1.5)).  The finalize kernel now FLAGS channels whose terms cancel
    below 1e-3 of their size and eml_dense_bn_dgamma_direct_f32 recomputes them as the f64 sum of dy * xhat.  Here: tiny,
    negative and zero gammas in BN1 of several layers and in a transition norm, gradients against the f64 oracle; with the
    fallback switched off (EML_DGAMMA_DIRECT=never) the same channels are far off."""
    anchors, crop, B = 32, (64, 96), 2
    ref32, net = _pair(anchors, crop, seed=7)
    sd = ref32.state_dict()
    g = np.random.default_rng(3)
    touched = {}
    for name in ("features.denseblock1.denselayer3.norm1.weight", "features.denseblock1.denselayer16.norm1.weight",
                 "features.denseblock2.denselayer8.norm1.weight", "features.denseblock3.denselayer2.norm1.weight",
                 "features.transition1.norm.weight", "features.transition3.norm.weight"):
        w = sd[name].clone()
        idx = g.choice(w.numel(), size=min(10, w.numel()), replace=False)
        vals = np.array([1e-5, -1e-5, 3e-4, -2e-3, 1e-7, 0.0, -0.7, 2e-6, -4e-4, 1e-3], dtype=np.float32)[:len(idx)]
        w[torch.from_numpy(idx)] = torch.from_numpy(vals)
        sd[name] = w
        touched[name] = idx[np.abs(vals) < 5e-3]          # the ill-conditioned ones
    ref64 = oracle.OracleDenseNet(anchors=anchors, crop_hw=crop).double()
    ref64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    ref64 = ref64.cuda().train()
    x = torch.from_numpy(g.random((B, 3) + crop, dtype=np.float32)).cuda()
    w = {k: torch.from_numpy(g.standard_normal(s).astype(np.float32)).cuda()
         for k, s in (("distribution", (B, anchors)), ("intensity", (B, 1)), ("rgb_ratio", (B, 3)), ("ambient", (B, 3)))}
    # two COMPLETED backwards with healthy gammas first -- round 3's "auto" mode read their "nothing
    # flagged" report and skipped the recomputation for the first steps after a channel crossed the threshold
    net.train()
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        o = net(x)
        sum((o[k] * w[k]).sum() for k in KEYS).backward()
        torch.cuda.synchronize()
    net.load_state_dict(sd)
    out = ref64(x.double())
    sum((out[k] * w[k].double()).sum() for k in KEYS).backward()
    truth = {n: q.grad.cpu().numpy() for n, q in ref64.named_parameters() if n in touched}

    def run(mode):
        monkeypatch.setenv("EML_DGAMMA_DIRECT", mode)
        net.zero_grad(set_to_none=True)
        o = net(x)
        sum((o[k] * w[k]).sum() for k in KEYS).backward()
        errs = []
        named = dict(net.named_parameters())
        for n, idx in touched.items():
            t = truth[n]
            rms = float(np.sqrt(np.mean(np.square(t))))
            errs.append(float(np.abs(named[n].grad.cpu().numpy().astype(np.float64)[idx] - t[idx]).max() / rms))
        return max(errs)
    e_auto = run("auto")      # the default: gated on THIS backward's flags, on the device
    e_always = run("always")
    e_never = run("never")
    print("small-gamma dgamma error / rms(grad): direct %.2e (auto) %.2e (always), identity only %.2e" % (e_auto, e_always, e_never))
    # the same bound the well-conditioned channels of this network meet against f64 (test_gradient_error_is_f32_conditioning)
    assert e_auto < 5e-2 and e_always < 5e-2
    assert e_never > 3 * max(e_auto, 1e-3), "the identity alone should be visibly worse on these channels (else the test is blind)"