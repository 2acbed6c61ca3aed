"""CPU: pin the oracle (oracle/) to vectors produced by the real reference
(tests/golden/make_golden.py).  Everything the GPU parity tests trust hangs off this."""
import numpy as np
import pytest
import torch

import oracle
from tests.golden.make_golden import SINKHORN_CASES

torch.set_num_threads(8)


def test_sphere_points_and_chord_matrix(golden_sinkhorn):
    for n in (96, 128):
        np.testing.assert_array_equal(oracle.sphere_points(n), golden_sinkhorn["sphere_points_%d" % n])
    np.testing.assert_allclose(oracle.anchor_cost_matrix(96).numpy(),
                               golden_sinkhorn["n96_blur025/M"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(oracle.anchor_cost_matrix(128).numpy(),
                               golden_sinkhorn["n128_blur05/M"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(oracle.anchor_cost_matrix(256).numpy()[::16],
                               golden_sinkhorn["M256_rows16"], rtol=0, atol=2e-7)


def test_cost_matrix(golden_sinkhorn):
    for name, n in (("n96_blur025", 96), ("n128_blur05", 128)):
        c = golden_sinkhorn.case(name)
        B = c["x"].shape[0]
        x = torch.from_numpy(c["x"]).view(B, n, 1)
        y = torch.from_numpy(c["y"]).view(B, n, 1)
        C = oracle.spherical_cost(x, y, torch.from_numpy(c["M"]))
        np.testing.assert_allclose(C.numpy(), c["C_xy"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("case", [c[0] for c in SINKHORN_CASES])
def test_samples_loss_matches_reference(golden_sinkhorn, case):
    c = golden_sinkhorn.case(case)
    B, n = c["x"].shape
    x = torch.from_numpy(c["x"]).view(B, n, 1).requires_grad_(True)
    y = torch.from_numpy(c["y"]).view(B, n, 1)
    M = oracle.anchor_cost_matrix(n)
    diam = None if c["fixed_diameter"] < 0 else float(c["fixed_diameter"])
    loss, aux = oracle.samples_loss(x, y, M, blur=float(c["blur"]), diameter=diam, return_aux=True)
    np.testing.assert_allclose(np.asarray(aux["eps_s"]), c["eps_s"], rtol=1e-12)
    assert abs(aux["diameter"] - float(c["diameter"])) <= 1e-7 * max(1.0, float(c["diameter"]))
    scale = max(1.0, float(np.abs(c["loss"]).max()) / 1e-4)
    np.testing.assert_allclose(loss.detach().numpy(), c["loss"], rtol=0, atol=1e-6 * scale)
    for got, want in zip(aux["duals"], c["duals"]):
        np.testing.assert_allclose(got.detach().numpy(), want, rtol=0, atol=1e-6 * max(1.0, np.abs(want).max()))
    loss.sum().backward()
    gref = c["grad_x"]
    np.testing.assert_allclose(x.grad.numpy().reshape(B, n), gref, rtol=1e-4, atol=2e-5 * np.abs(gref).max())
    # analytic gradient formula (what the HIP backward implements) == reference autograd
    ga = oracle.samples_loss_grad_analytic(x.detach(), y, M, aux["eps_s"]).numpy().reshape(B, n)
    np.testing.assert_allclose(ga, gref, rtol=1e-4, atol=2e-5 * np.abs(gref).max())


@pytest.mark.parametrize("case,height", [("b1_n128", 128), ("b2_n96", 128), ("b3_n42", 128), ("lat256", 256)])
def test_rasteriser_matches_reference(golden_raster, case, height):
    c = golden_raster.case(case)
    pano = oracle.convert_to_panorama(torch.from_numpy(c["dirs"]), torch.from_numpy(c["sizes"]),
                                      torch.from_numpy(c["colors"]), height=height).numpy()
    rows = int(c["row_stride"])
    want = c["pano_rows"]
    np.testing.assert_allclose(pano[:, :, ::rows], want, rtol=1e-5, atol=1e-6 * want.max())
    assert abs(pano.astype(np.float64).sum() - float(c["sum"])) <= 1e-5 * abs(float(c["sum"]))


def _sample(t, idx):
    return t.detach().reshape(-1)[torch.from_numpy(idx)].numpy()


def test_densenet_eval_and_train_step(golden_densenet):
    g = golden_densenet
    net = oracle.OracleDenseNet()
    net.load_state_dict(oracle.deterministic_state_dict(net.state_dict(), seed=0))
    assert sum(p.numel() for p in net.parameters()) == 9336711
    x = torch.from_numpy(np.random.default_rng([0]).random((2, 3, 192, 256), dtype=np.float32))
    net.eval()
    with torch.no_grad():
        pe, fe = net(x), net.features_forward(x)
    for k in ("distribution", "intensity", "rgb_ratio", "ambient"):
        np.testing.assert_allclose(pe[k].numpy(), g["eval/" + k], rtol=0, atol=1e-5)
    np.testing.assert_allclose(_sample(fe, g["eval/features_idx"]), g["eval/features_sample"], rtol=0, atol=1e-5)

    net.train()
    gt = {k: torch.from_numpy(g["train/gt_" + k]) for k in ("distribution", "intensity", "rgb_ratio", "ambient")}
    M = oracle.anchor_cost_matrix(96)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    pred = net(x)
    loss, terms = oracle.regression_loss(pred, gt, lambda a, b: oracle.samples_loss(a, b, M, blur=.025), 96)
    opt.zero_grad()
    loss.backward()
    for k in ("distribution", "intensity", "rgb_ratio", "ambient"):
        np.testing.assert_allclose(pred[k].detach().numpy(), g["train/" + k], rtol=0, atol=1e-4)
    np.testing.assert_allclose(np.array([float(t) for t in terms.values()]), g["train/loss_terms"], rtol=1e-4)
    named = dict(net.named_parameters())
    for key in [k[len("train/grad/"):] for k in g.z.files if k.startswith("train/grad/")]:
        got = _sample(named[key].grad, g["train/grad_idx/" + key])
        want = g["train/grad/" + key]
        l2 = float(g["train/grad_l2/" + key])
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-4 * l2 / np.sqrt(named[key].numel()) + 1e-7)
    np.testing.assert_allclose(net.features.norm0.running_mean.numpy(), g["train/running_mean/features.norm0"], atol=1e-6)
    np.testing.assert_allclose(net.features.last_norm3.running_var.numpy(), g["train/running_var/features.last_norm3"], rtol=1e-4)
    opt.step()
    np.testing.assert_allclose(net.fc_dist.bias.detach().numpy(), g["train/post_step/fc_dist.bias"], rtol=0, atol=2e-6)


def test_densenet_cfg1_240x320_128_anchors(golden_densenet):
    """BASELINE cfg1: 1x240x320 crop -> 128 anchors, CPU forward only."""
    g = golden_densenet
    net = oracle.OracleDenseNet(anchors=128, crop_hw=(240, 320))
    net.load_state_dict(oracle.deterministic_state_dict(net.state_dict(), seed=1))
    net.eval()
    x = torch.from_numpy(np.random.default_rng([2]).random((1, 3, 240, 320), dtype=np.float32))
    with torch.no_grad():
        p = net(x)
    assert p["distribution"].shape == (1, 128)
    for k in ("distribution", "intensity", "rgb_ratio", "ambient"):
        np.testing.assert_allclose(p[k].numpy(), g["cfg1/" + k], rtol=0, atol=1e-5)


def test_densenet_train_step_baseline_geometry(golden_densenet_cfg2):
    """One full training step of the reference at BASELINE's geometry (240x320 crops, 128 anchors, blur .05, B = 2;
    ``make_golden.gen_densenet_cfg2``: the reference class with fc / fc_dist swapped): predictions, the five
    loss terms, 17 sampled gradient tensors (block 1 at 240x320 included), running statistics, post-Adam bias."""
    g = golden_densenet_cfg2
    net = oracle.OracleDenseNet(anchors=128, crop_hw=(240, 320))
    net.load_state_dict(oracle.deterministic_state_dict(net.state_dict(), seed=5))
    net.train()
    x = torch.from_numpy(np.random.default_rng([40]).random((2, 3, 240, 320), dtype=np.float32))
    gt = {k: torch.from_numpy(g["train/gt_" + k]) for k in ("distribution", "intensity", "rgb_ratio", "ambient")}
    M = oracle.anchor_cost_matrix(128)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    pred = net(x)
    loss, terms = oracle.regression_loss(pred, gt, lambda a, b: oracle.samples_loss(a, b, M, blur=.05), 128)
    opt.zero_grad()
    loss.backward()
    for k in ("distribution", "intensity", "rgb_ratio", "ambient"):
        np.testing.assert_allclose(pred[k].detach().numpy(), g["train/" + k], rtol=0, atol=1e-4)
    np.testing.assert_allclose(np.array([float(t.detach()) for t in terms.values()]), g["train/loss_terms"], rtol=1e-4)
    named = dict(net.named_parameters())
    for key in [k[len("train/grad/"):] for k in g.z.files if k.startswith("train/grad/")]:
        got = _sample(named[key].grad, g["train/grad_idx/" + key])
        want = g["train/grad/" + key]
        l2 = float(g["train/grad_l2/" + key])
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-4 * l2 / np.sqrt(named[key].numel()) + 1e-7, err_msg=key)
    np.testing.assert_allclose(net.features.norm0.running_mean.numpy(), g["train/running_mean/features.norm0"], atol=1e-6)
    np.testing.assert_allclose(net.features.denseblock2.denselayer3.norm2.running_var.numpy(),
                               g["train/running_var/features.denseblock2.denselayer3.norm2"], rtol=1e-4)
    np.testing.assert_allclose(net.features.last_norm3.running_var.numpy(), g["train/running_var/features.last_norm3"], rtol=1e-4)
    opt.step()
    np.testing.assert_allclose(net.fc_dist.bias.detach().numpy(), g["train/post_step/fc_dist.bias"], rtol=0, atol=2e-6)


def test_gmloss_geometry_cost_matches_reference():
    """GMLight variant (RegressionNetwork/gmloss): depth-scaled anchors, per-call chord matrix, same Sinkhorn."""
    from tests.conftest import Golden
    from tests.golden.make_golden import gmloss_inputs
    g = Golden("gmloss")
    for name, B in (("b3_blur05", 3), ("b2_blur025", 2)):
        x_np, y_np, depth = gmloss_inputs(B, 17)
        anchors = oracle.geometric_points(128, depth)
        np.testing.assert_allclose(anchors.astype(np.float32), g[name + "/anchors"], rtol=0, atol=1e-7)
        M = oracle.cost_matrix_of(anchors)
        np.testing.assert_allclose(M[::8].numpy(), g[name + "/M_rows8"], rtol=0, atol=1e-6)
        x = torch.from_numpy(x_np).view(B, 128, 1).requires_grad_(True)
        y = torch.from_numpy(y_np).view(B, 128, 1)
        loss = oracle.samples_loss(x, y, M, blur=float(g[name + "/blur"]))
        np.testing.assert_allclose(loss.detach().numpy(), g[name + "/loss"], rtol=1e-5, atol=1e-7)
        loss.sum().backward()
        np.testing.assert_allclose(x.grad.numpy().reshape(B, 128), g[name + "/grad_x"], rtol=1e-4, atol=1e-8)


def test_gt_parametrisation_matches_reference():
    """extract_mesh (representation/distribution_representation.py:65-120): nearest-anchor map and the four parameter
    groups of synthetic HDR panoramas against the reference class itself."""
    from tests.conftest import Golden
    from tests.golden.make_golden import gt_hdr_inputs
    g = Golden("gt_param")
    for name, h, w, ln, B in (("h128_n128", 128, 256, 128, 2), ("h64_n96", 64, 128, 96, 2)):
        ex = oracle.ExtractMesh(h=h, w=w, ln=ln)
        np.testing.assert_array_equal(ex.idx.astype(np.int32), g[name + "/idx"])
        hdr = gt_hdr_inputs(B, h, w, 5)
        for b in range(B):
            para, mp = ex.compute(hdr[b])
            assert int(mp.sum()) == int(g["%s/%d/map_count" % (name, b)])
            for k in ("distribution", "intensity", "rgb_ratio", "ambient"):
                np.testing.assert_allclose(para[k], g["%s/%d/%s" % (name, b, k)], rtol=1e-12, atol=1e-14, err_msg=k)
