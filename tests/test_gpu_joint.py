"""GPU parity of the joint regression -> rasteriser -> projector step (BASELINE configs[3], emlight_amd/joint.py).

Small size: one full iteration of the HIP path (DenseNet engine, Sinkhorn, SG rasteriser forward + colour gradient,
SphereConv2D / SPADE kernels, three Adam updates) against the same composition built from oracle pieces on the CPU
(oracle.joint_step: OracleDenseNet, oracle Sinkhorn, oracle rasteriser, the projector modules on stock ops).
Full size (32 per GPU, 240x320 crops, ngf = ndf = 64): size-independent properties."""
import copy

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
KEYS = ("distribution", "intensity", "rgb_ratio", "ambient")


def _rel_l2(a, b, floor):
    rms = lambda t: float(np.sqrt(np.mean(np.square(t))))
    return rms(a - b) / max(rms(b), floor)


def test_joint_step_small_vs_oracle_composition():
    from emlight_amd.GenProjector import networks
    from emlight_amd.GenProjector.pix2pix_model import Pix2PixModel
    from emlight_amd.joint import JointTrainer, joint_batch
    from emlight_amd.GenProjector.vgg import VGG19Features
    ln, crop, B = 128, (64, 96), 2
    # the configuration bench.py and the CLIs time: VGG perceptual term ON, the same seeded
    # torchvision-style state dict on both sides (HIP gather-GEMM stack vs stock nn.Sequential)
    vgg_sd = oracle.seeded_vgg19_state_dict(seed=3)
    opt = networks.default_options(ngf=8, ndf=8, no_vgg_loss=False)
    tr = JointTrainer(opt, anchors=ln, crop_hw=crop, blur=.05, device="cuda:0", vgg_features=VGG19Features(state_dict=vgg_sd))
    enc_o = oracle.OracleDenseNet(anchors=ln, crop_hw=crop).train()
    sd = oracle.deterministic_state_dict(enc_o.state_dict(), seed=21)
    enc_o.load_state_dict(sd)
    tr.reg.model.load_state_dict(sd)
    pm_o = Pix2PixModel(opt, vgg_features=oracle.StockVGG19(vgg_sd)).train()
    sdG = oracle.deterministic_projector_state_dict(pm_o.netG.state_dict(), seed=11)
    sdD = oracle.deterministic_projector_state_dict(pm_o.netD.state_dict(), seed=12)
    for m in (pm_o, tr.proj.model):
        m.netG.load_state_dict(sdG)
        m.netD.load_state_dict(sdD)
    batch = joint_batch(B, "cuda:0", ln, crop, seed=77)
    batch_cpu = {k: v.cpu() for k, v in batch.items()}

    # The iteration runs in its two halves (JointTrainer.step == generator_step + discriminator_step).  The D half is
    # evaluated AFTER Adam's first update of G, which moves every entry by lr * sign(g): two f32 evaluations disagree on
    # the sign where g ~ 0, so comparing the D half of two independently updated generators measures that, not the D-path
    # kernels.  The halves are therefore decoupled: after the first half the oracle's UPDATED generator is copied into the
    # product, and the D half is compared from identical weights at the same bounds as the generator's.
    data = tr.generator_step(batch)
    got = dict(tr.losses)
    M = oracle.anchor_cost_matrix(ln)
    emd = lambda a, b: oracle.samples_loss(a, b, M, blur=.05)
    opt_E = torch.optim.Adam(enc_o.parameters(), lr=1e-4, betas=(0.9, 0.999))
    opt_G, opt_D = pm_o.create_optimizers(opt)
    with oracle.stock_sphere_ops():
        want = oracle.joint_generator_step(enc_o, pm_o, opt_E, opt_G, batch_cpu, emd, ln)

    # forward: guide map (HDR map: north_star 1e-2 rel, held to 1e-4 of peak), generated panorama, every loss term
    gm = want["gmap"].detach().numpy()
    np.testing.assert_allclose(tr.guide.detach().cpu().numpy(), gm, rtol=1e-4, atol=1e-4 * np.abs(gm).max())
    fk = want["fake"].detach().numpy()
    np.testing.assert_allclose(tr.generated.detach().cpu().numpy(), fk, rtol=1e-3, atol=2e-3)
    ref_losses = {**want["terms"], **want["g_losses"]}
    assert set(got) == set(ref_losses) and "VGG" in got
    for k, v in ref_losses.items():
        np.testing.assert_allclose(float(got[k].detach().mean()), float(v.detach().mean()), rtol=1e-4, atol=1e-6, err_msg=k)

    # backward: relative L2 per tensor (element-wise agreement is impossible across two f32 ReLU networks)
    def check(named_got, named_want, max_bound, med_bound, what):
        grads = {k: p.grad.numpy() for k, p in named_want.items() if p.grad is not None}
        floor = 1e-3 * float(np.median([np.sqrt(np.mean(np.square(g))) for g in grads.values()]))
        errs = sorted(((_rel_l2(named_got[k].grad.cpu().numpy(), g, floor), k) for k, g in grads.items()), reverse=True)
        assert all(np.isfinite(e) for e, _ in errs), what
        print("%s gradients: worst %s, median %.2e" % (what, errs[:3], np.median([e for e, _ in errs])))
        assert errs[0][0] < max_bound, "%s: largest relative L2 grad errors %s" % (what, errs[:6])
        assert np.median([e for e, _ in errs]) < med_bound, "%s: median %g" % (what, np.median([e for e, _ in errs]))
    # the stand-alone encoder meets 2e-2 (tests/test_gpu_densenet.py); so does the one driven through the rasteriser
    check(dict(tr.reg.model.named_parameters()), dict(enc_o.named_parameters()), 2e-2, 5e-3, "encoder")
    check(dict(tr.proj.model.netG.named_parameters()), dict(pm_o.netG.named_parameters()), 2e-2, 2e-3, "generator")

    # ---- D half from IDENTICAL generator weights (the oracle's post-Adam G, spectral-norm vectors and BN buffers included)
    tr.proj.model.netG.load_state_dict(pm_o.netG.state_dict())
    got_d = tr.discriminator_step(data)
    with oracle.stock_sphere_ops():
        want_d = oracle.joint_discriminator_step(pm_o, opt_D, want["data"])
    assert set(got_d) == set(want_d)
    for k, v in want_d.items():
        np.testing.assert_allclose(float(got_d[k].detach().mean()), float(v.detach().mean()), rtol=1e-3, atol=1e-6, err_msg=k)
    check(dict(tr.proj.model.netD.named_parameters()), dict(pm_o.netD.named_parameters()), 2e-2, 2e-3, "discriminator")

    # the projector's losses really reach the encoder through the rasteriser: the regression-only gradient differs
    enc_r = oracle.OracleDenseNet(anchors=ln, crop_hw=crop).train()
    enc_r.load_state_dict(sd)
    l_reg, _ = oracle.regression_loss(enc_r(batch_cpu["crop"]), batch_cpu, emd, ln)
    l_reg.backward()
    d = (enc_r.fc_intensity.weight.grad - enc_o.fc_intensity.weight.grad).abs().max()
    assert float(d) > 1e-6 * float(enc_r.fc_intensity.weight.grad.abs().max())


def test_joint_step_properties_at_cfg4_size():
    """32 images per GPU (BASELINE configs[3]: 256 over 8), 240x320 crops, 128 anchors, ngf = ndf = 64: too large for
    the CPU oracle, so: (1) run-to-run reproducibility of two whole iterations (the HIP kernels use no atomics and are
    bitwise reproducible on their own; the library GEMM / convolution calls between them may reorder sums, so: 1e-3);
    (2) additivity of the backward -- the encoder gradient of L_reg + L_G equals grad(L_reg) + grad(L_G);
    (3) every loss finite, guide map == rasteriser of the predicted parameters + ambient."""
    from emlight_amd.GenProjector import networks
    from emlight_amd.joint import JointTrainer, joint_batch, predicted_gaussian_map
    from emlight_amd.RegressionNetwork.engine import regression_loss
    dev, ln, crop, B = "cuda:0", 128, (240, 320), 32
    batch = joint_batch(B, dev, ln, crop, seed=5)

    def snapshot(tr):
        mods = (tr.reg.model, tr.proj.model)
        opts = (tr.reg.optimizer, tr.proj.optimizer_G, tr.proj.optimizer_D)
        return [copy.deepcopy(m.state_dict()) for m in mods], [copy.deepcopy(o.state_dict()) for o in opts]

    def restore(tr, snap):
        for m, sd in zip((tr.reg.model, tr.proj.model), snap[0]):
            m.load_state_dict(sd)
        for o, sd in zip((tr.reg.optimizer, tr.proj.optimizer_G, tr.proj.optimizer_D), snap[1]):
            o.load_state_dict(copy.deepcopy(sd))

    def one(tr):
        return {k: v.detach().clone() for k, v in tr.step(batch).items()}

    def same(la, lb, what):
        # GAN / D terms are O(1e-2) differences of O(1) discriminator outputs, hence the absolute part
        for k in la:
            assert torch.isfinite(la[k]).all(), k
            torch.testing.assert_close(la[k], lb[k], rtol=1e-3, atol=2e-4,
                                       msg=lambda m, k=k: "%s must be reproducible (%s): %s" % (what, k, m))
    torch.manual_seed(3)
    tr = JointTrainer(networks.default_options(), anchors=ln, crop_hw=crop, blur=.05, device=dev)
    s0 = snapshot(tr)
    a1 = one(tr)
    # Iteration 2 is compared from ONE post-update state replayed twice: two independently trained copies differ after
    # Adam's first update by lr * sign(g) wherever g ~ 0 (that is the optimiser's conditioning, not the kernels'), which
    # would force a 10 % bound; from identical weights + optimiser state the second iteration holds the first's bound.
    s1 = snapshot(tr)
    a2 = one(tr)
    enc_w = tr.reg.model.fc_dist.weight.detach().clone()
    g_w = tr.proj.model.netG.sphere_conv1.weight.detach().clone()
    restore(tr, s1)
    b2 = one(tr)
    same(a2, b2, "joint iteration 2 (replayed from the state after iteration 1)")
    # the replayed Adam step lands on the same weights up to the sign of entries with g ~ 0 (at most 2 * lr apart)
    for a_w, b_w in ((enc_w, tr.reg.model.fc_dist.weight.detach()), (g_w, tr.proj.model.netG.sphere_conv1.weight.detach())):
        d = (a_w - b_w).abs()
        assert float(d.mean()) < 2e-5 and float(d.max()) <= 4.1e-4, (float(d.mean()), float(d.max()))
    restore(tr, s0)
    same(a1, one(tr), "joint iteration 1 (replayed from the initial state)")
    del s0, s1

    # (2) additivity on the encoder, no optimiser steps
    enc, pm = tr.reg.model, tr.proj.model
    state = copy.deepcopy(enc.state_dict())

    def enc_grads(use_reg, use_g):
        enc.load_state_dict(state)
        enc.zero_grad(set_to_none=True)
        pm.zero_grad(set_to_none=True)
        pred = enc(batch["crop"])
        l_reg, _ = regression_loss(pred, batch, tr.reg.sam_loss, ln)
        data = tr.projector_inputs(batch, pred)
        g_losses, _ = pm(data, mode="generator")
        total = (l_reg if use_reg else 0.0) + (sum(g_losses.values()).mean() if use_g else 0.0)
        total.backward()
        return [p.grad.clone() for p in enc.parameters()], data["input"].detach(), pred
    pm.eval()   # freeze spectral-norm power iterations / BN statistics between the three passes
    both, gmap, pred = enc_grads(True, True)
    reg_only, _, _ = enc_grads(True, False)
    g_only, _, _ = enc_grads(False, True)
    # exact in real arithmetic; in f32 the two upstream gradients take different rounding paths through ~100 layers
    # (measured 2.5e-3 of the tensor's largest entry, the conditioning of section 4 of DESIGN.md)
    rms = lambda t: float(t.double().square().mean().sqrt())
    floor = 1e-2 * float(np.median([rms(g) for g in both]))   # tensors whose gradient nearly cancels: absolute scale
    worst = max(rms(gb - (gr + gg)) / max(rms(gb), floor) for gb, gr, gg in zip(both, reg_only, g_only))
    assert worst <= 2e-2, worst
    assert any(float(g.abs().max()) > 0 for g in g_only), "the generator losses must reach the encoder"
    want = predicted_gaussian_map({k: v.detach() for k, v in pred.items()}, ln)
    assert torch.equal(want, gmap)   # the rasteriser itself is bitwise reproducible


def test_two_iterations_are_bit_reproducible():
    """Run-to-run: two trainers from the same seed, two joint iterations each -- every loss term and every parameter of the
    encoder, the generator and the discriminator must agree BIT FOR BIT.  The HIP kernels of this package use no atomics and no
    timing-dependent split; the only run-to-run differences of a joint step come from the library's planar convolutions of the
    generator's crop encoder (``nn.Conv2d`` -> MIOpen, whose default weight-gradient algorithm sums with atomics: measured
    spread 5e-6 on ``GAN_Feat`` over four runs, tools/exp/ddp_flake_probe.py), so MIOpen is asked for its deterministic
    algorithms here (``torch.backends.cudnn.deterministic``).  With that, eight fresh processes gave identical numbers."""
    from emlight_amd.GenProjector import networks
    from emlight_amd.joint import JointTrainer, joint_batch
    before = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        runs = []
        for _ in range(2):
            torch.manual_seed(0)
            tr = JointTrainer(networks.default_options(ngf=4, ndf=4), anchors=32, crop_hw=(64, 96), device="cuda:0")
            batch = joint_batch(2, "cuda:0", 32, (64, 96), seed=9)
            for _ in range(2):
                losses = tr.step(batch)
            nets = (tr.reg.model, tr.proj.model.netG, tr.proj.model.netD)
            runs.append(({k: v.detach().clone() for k, v in losses.items()},
                         [(n, q.detach().clone()) for net in nets for n, q in net.named_parameters()]))
            del tr
    finally:
        torch.backends.cudnn.deterministic = before
    for k in runs[0][0]:
        assert torch.equal(runs[0][0][k], runs[1][0][k]), (k, runs[0][0][k], runs[1][0][k])
    for (n, a), (_, b) in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b), (n, float((a - b).abs().max()))
