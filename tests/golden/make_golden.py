#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference on seeded inputs (CPU).

Run only in the build container, where ``/root/reference`` exists:

    python tests/golden/make_golden.py

It imports the reference's own modules (never copied into this repo), feeds them
numpy-seeded inputs, and stores inputs' seeds + outputs as small ``.npz`` vectors.
The GPU box has no ``/root/reference``; tests there read only the committed vectors.

Import shims: ``Tensor.cuda``/``Module.cuda`` become
no-ops (the reference calls ``.cuda()`` unconditionally, ``geomloss/utils.py:80-81``,
``GenProjector/util.py:357``), and absent third-party modules that the reference
imports at file top but that the hot path never touches (cv2, OpenEXR, Imath,
imageio, vtk) are stubbed with ``MagicMock``.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))



def install_shims():
    """Only when generating (never on import: the tests import SINKHORN_CASES from here)."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.set_num_threads(8)


def rng(*seed):
    return np.random.default_rng(list(seed))


def softmax_np(v, axis=-1):
    v = v - v.max(axis=axis, keepdims=True)
    e = np.exp(v)
    return (e / e.sum(axis=axis, keepdims=True)).astype(np.float32)


# --------------------------------------------------------------------------- sinkhorn
def ref_geomloss():
    sys.path.insert(0, os.path.join(REF, "RegressionNetwork"))
    import geomloss  # noqa: the reference package
    from geomloss import utils as gutils
    return geomloss, gutils


def ref_samples_loss(geomloss, gutils, n, batch, blur, diameter=None):
    """A reference SamplesLoss whose `distance` has N == n.

    n == 96 uses the reference constructor verbatim.  For other n the reference
    hard-codes N (utils.py:66), so the `distance` object is allocated without running
    its __init__ and given M built by the reference's own recipe (sphere_points ->
    f32 -> pairwise torch.norm, utils.py:67-76); every other line that runs
    (spherical_distance, sinkhorn_loop, softmin_tensorized, ...) is the reference's.
    """
    if n == 96:
        return geomloss.SamplesLoss("sinkhorn", p=2, blur=blur, diameter=diameter, batchsize=batch)
    anchors = torch.from_numpy(gutils.sphere_points(n)).float()
    M = torch.ones((n, n))
    for i in range(n):
        M[i] = torch.norm(anchors[i][None, :] - anchors, dim=1)
        for j in (0, n // 3, n - 1):  # spot-check the vectorised row against the scalar form
            assert M[i, j] == torch.norm(anchors[i] - anchors[j])
    M[M < 0] = 0
    d = gutils.distance.__new__(gutils.distance)
    d.N = n
    d.anchors = anchors.unsqueeze(0).repeat(batch, 1, 1)
    d.M = M.unsqueeze(0).repeat(batch, 1, 1)
    loss = geomloss.SamplesLoss.__new__(geomloss.SamplesLoss)
    torch.nn.Module.__init__(loss)
    loss.loss, loss.p, loss.blur, loss.reach = "sinkhorn", 2, blur, None
    loss.diameter, loss.scaling, loss.distance = diameter, .5, d
    return loss


def sinkhorn_inputs(kind, B, n, seed):
    g = rng(seed, n, B)
    if kind == "softmax":
        x = softmax_np(g.standard_normal((B, n)))
        y = softmax_np(2.0 * g.standard_normal((B, n)))
    elif kind == "sparse":  # GT-like: sparse simplex target
        x = softmax_np(g.standard_normal((B, n)))
        y = softmax_np(4.0 * g.standard_normal((B, n)))
        thr = np.quantile(y, 0.75, axis=1, keepdims=True)
        y = np.where(y < thr, 0.0, y)
        y = (y / y.sum(1, keepdims=True)).astype(np.float32)
    elif kind == "logits":
        x = g.standard_normal((B, n)).astype(np.float32)
        y = g.standard_normal((B, n)).astype(np.float32)
    elif kind == "tiny":  # diameter < blur: schedule collapses to 2 entries
        x = (0.01 + 1e-3 * g.random((B, n))).astype(np.float32)
        y = (0.01 + 1e-3 * g.random((B, n))).astype(np.float32)
    else:
        raise ValueError(kind)
    return x, y


SINKHORN_CASES = [
    # name, kind, B, N, blur, diameter
    ("n96_blur025", "softmax", 4, 96, .025, None),   # train.py:61 setting
    ("n96_blur05", "softmax", 4, 96, .05, None),     # SamplesLoss default
    ("n96_sparse", "sparse", 4, 96, .05, None),
    ("n96_logits", "logits", 3, 96, .025, None),     # raw +-3 sigma logits: ~12 eps steps
    ("n96_tiny", "tiny", 2, 96, .05, None),          # diameter < blur edge
    ("n96_fixdiam", "softmax", 4, 96, .05, 1.0),
    ("n128_blur05", "sparse", 4, 128, .05, None),    # BASELINE cfg2 shape
    ("n128_blur025", "softmax", 2, 128, .025, None),
    ("n256_blur05", "sparse", 2, 256, .05, None),    # BASELINE cfg5 shape
    ("n64_blur05", "softmax", 3, 64, .05, None),
    ("n50_blur05", "softmax", 2, 50, .05, None),     # ragged N (not a multiple of 32)
]


def gen_sinkhorn():
    geomloss, gutils = ref_geomloss()
    from geomloss import sinkhorn_divergence as sd
    out = {}
    for name, kind, B, n, blur, diam in SINKHORN_CASES:
        x_np, y_np = sinkhorn_inputs(kind, B, n, 7)
        x = torch.from_numpy(x_np).view(B, n, 1).requires_grad_(True)
        y = torch.from_numpy(y_np).view(B, n, 1)
        crit = ref_samples_loss(geomloss, gutils, n, B, blur, diam)
        # capture the schedule and the duals the reference computes internally
        cap = {}
        orig_sp, orig_cost = sd.scaling_parameters, sd.sinkhorn_cost
        import geomloss.samples_loss as sl

        def sp(*a, **k):
            r = orig_sp(*a, **k)
            cap["diameter"], cap["eps_s"] = r[0], list(r[2])
            return r

        def sc(eps, rho, a, b, a_x, b_y, a_y, b_x):
            cap["duals"] = [t.detach().numpy().copy() for t in (a_x, b_y, a_y, b_x)]
            return orig_cost(eps, rho, a, b, a_x, b_y, a_y, b_x)

        sl.scaling_parameters, sl.sinkhorn_cost = sp, sc
        try:
            loss = crit(x, y)
        finally:
            sl.scaling_parameters, sl.sinkhorn_cost = orig_sp, orig_cost
        loss.sum().backward()
        out[name + "/x"] = x_np
        out[name + "/y"] = y_np
        out[name + "/loss"] = loss.detach().numpy()
        out[name + "/grad_x"] = x.grad.numpy().reshape(B, n)
        out[name + "/eps_s"] = np.asarray(cap["eps_s"], dtype=np.float64)
        out[name + "/diameter"] = np.float64(cap["diameter"])
        out[name + "/duals"] = np.stack(cap["duals"])
        out[name + "/blur"] = np.float64(blur)
        out[name + "/fixed_diameter"] = np.float64(-1.0 if diam is None else diam)
        if name in ("n96_blur025", "n128_blur05"):
            out[name + "/M"] = crit.distance.M[0].numpy()
            C = crit.distance.spherical_distance(x.detach(), y) / 2
            out[name + "/C_xy"] = C.numpy()
        print("sinkhorn", name, "n_eps", len(cap["eps_s"]), "loss0 %.4e" % float(loss[0]))
    # reference M checksums for N=256 (full matrix is 256 KB: store every 16th row)
    crit = ref_samples_loss(geomloss, gutils, 256, 1, .05)
    out["M256_rows16"] = crit.distance.M[0, ::16].numpy()
    out["sphere_points_96"] = gutils.sphere_points(96)
    out["sphere_points_128"] = gutils.sphere_points(128)
    np.savez_compressed(os.path.join(HERE, "sinkhorn.npz"), **out)


# --------------------------------------------------------------------------- gmloss (GMLight geometry cost)
def gmloss_inputs(B, seed):
    """x, y on the 128 anchors plus a per-anchor depth (``gmloss/utils.py:63-74``: radius = anchor_depth)."""
    x, y = sinkhorn_inputs("softmax", B, 128, seed)
    depth = rng(seed, 128).uniform(0.5, 3.0, 128)
    return x, y, depth


def gen_gmloss():
    """``RegressionNetwork/gmloss``: SamplesLoss.forward(x, y, geometry) -- the chord matrix of depth-scaled anchors is
    rebuilt per call (``gmloss/samples_loss.py:72``, an N^2 Python loop in ``gmloss/utils.py:76-92``)."""
    sys.path.insert(0, os.path.join(REF, "RegressionNetwork"))
    import gmloss  # noqa: the reference package
    out = {}
    for name, B, blur in (("b3_blur05", 3, .05), ("b2_blur025", 2, .025)):
        x_np, y_np, depth = gmloss_inputs(B, 17)
        x = torch.from_numpy(x_np).view(B, 128, 1).requires_grad_(True)
        y = torch.from_numpy(y_np).view(B, 128, 1)
        crit = gmloss.SamplesLoss("sinkhorn", p=2, blur=blur, batchsize=B)
        loss = crit(x, y, depth)
        loss.sum().backward()
        out[name + "/loss"] = loss.detach().numpy()
        out[name + "/grad_x"] = x.grad.numpy().reshape(B, 128)
        out[name + "/M_rows8"] = crit.distance.M[0, ::8].numpy()
        out[name + "/anchors"] = crit.distance.anchors[0].numpy()
        out[name + "/blur"] = np.float64(blur)
        print("gmloss", name, "loss0 %.4e" % float(loss[0]))
    np.savez_compressed(os.path.join(HERE, "gmloss.npz"), **out)


# --------------------------------------------------------------------------- GT parametrisation
def gt_hdr_inputs(B, h, w, seed):
    """Synthetic HDR panoramas (B, h, w, 3) f32: a few bright lobes on a dim, smooth ambient field."""
    g = rng(seed, B, h, w)
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    out = np.zeros((B, h, w, 3), dtype=np.float64)
    for b in range(B):
        out[b] = 0.05 * (1 + np.sin(6 * xx + b)[..., None] * g.uniform(0.1, 0.3, 3))
        for _ in range(3):
            cy, cx, s = g.uniform(0.1, 0.9), g.uniform(0.05, 0.95), g.uniform(0.01, 0.04)
            out[b] += (np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)) * g.uniform(20, 200))[..., None] * g.uniform(0.5, 1.0, 3)
    return out.astype(np.float32)


def ref_extract_mesh():
    """The reference class itself: ``distribution_representation.py`` runs a dataset loop at import time, so only its
    ``extract_mesh`` ClassDef is compiled (from the file where it lies) against the reference's own ``util`` module."""
    import ast
    stub_io_modules()
    sys.modules.setdefault("detect_util", MagicMock())
    rep = os.path.join(REF, "RegressionNetwork", "representation")
    sys.path.insert(0, rep)
    import importlib.util as ilu
    spec = ilu.spec_from_file_location("ref_representation_util", os.path.join(rep, "util.py"))
    util = ilu.module_from_spec(spec)
    spec.loader.exec_module(util)
    src = open(os.path.join(rep, "distribution_representation.py")).read()
    node = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "extract_mesh"][0]
    ns = {"np": np, "util": util}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "distribution_representation.py", "exec"), ns)
    return ns["extract_mesh"]


def gen_gt_param():
    cls = ref_extract_mesh()
    out = {}
    for name, h, w, ln, B in (("h128_n128", 128, 256, 128, 2), ("h64_n96", 64, 128, 96, 2)):
        ex = cls(h=h, w=w, ln=ln)
        hdr = gt_hdr_inputs(B, h, w, 5)
        out[name + "/idx"] = ex.idx.astype(np.int32)
        for b in range(B):
            para, mp = ex.compute(hdr[b])
            for k in ("distribution", "intensity", "rgb_ratio", "ambient"):
                out["%s/%d/%s" % (name, b, k)] = np.asarray(para[k], dtype=np.float64)
            out["%s/%d/map_count" % (name, b)] = np.int64(mp.sum())
        print("gt_param", name, "intensity %.4f" % float(para["intensity"]), "lit pixels", int(mp.sum()))
    np.savez_compressed(os.path.join(HERE, "gt_param.npz"), **out)


# --------------------------------------------------------------------------- rasteriser
def stub_io_modules():
    for m in ["cv2", "OpenEXR", "Imath", "imageio", "imageio.plugins",
              "imageio.plugins.freeimage", "vtk", "vtk.util", "vtk.util.numpy_support",
              "torchvision", "torchvision.transforms", "torchvision.models",
              "matplotlib", "matplotlib.pyplot"]:
        sys.modules.setdefault(m, MagicMock())


def raster_inputs(B, n, seed, anchors):
    g = rng(seed, B, n)
    dirs = np.tile(anchors.reshape(1, 3 * n), (B, 1)).astype(np.float32)
    sizes = np.full((B, n), 0.0025, dtype=np.float32)
    dist = softmax_np(3.0 * g.standard_normal((B, n)))
    inten = g.uniform(50, 500, (B, 1, 1))
    rgb = g.uniform(0.4, 0.7, (B, 1, 3))
    colors = (dist[:, :, None] * inten * rgb).reshape(B, 3 * n).astype(np.float32)
    return dirs, sizes, colors


def gen_rasteriser():
    stub_io_modules()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gp_util", os.path.join(REF, "GenProjector", "util.py"))
    gp_util = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gp_util)
    out = {}
    _, gutils = ref_geomloss()
    for name, B, n, rows in [("b1_n128", 1, 128, 1), ("b2_n96", 2, 96, 4), ("b3_n42", 3, 42, 8)]:
        dirs, sizes, colors = raster_inputs(B, n, 11, gutils.sphere_points(n))
        if name == "b3_n42":  # varied lobe widths + off-anchor directions
            g = rng(5)
            sizes = g.uniform(0.002, 0.3, sizes.shape).astype(np.float32)
            d = g.standard_normal((B, n, 3))
            dirs = (d / np.linalg.norm(d, axis=2, keepdims=True)).reshape(B, 3 * n).astype(np.float32)
        pano = gp_util.convert_to_panorama(torch.from_numpy(dirs), torch.from_numpy(sizes),
                                           torch.from_numpy(colors)).numpy()
        out[name + "/dirs"], out[name + "/sizes"], out[name + "/colors"] = dirs, sizes, colors
        out[name + "/pano_rows"] = pano[:, :, ::rows]
        out[name + "/row_stride"] = np.int64(rows)
        out[name + "/sum"] = np.float64(pano.astype(np.float64).sum())
        print("raster", name, pano.shape, "sum %.3f max %.4f" % (pano.sum(), pano.max()))
    # variable-latitude form (panorama.py:68-82,142-152) at 256x512 -- BASELINE cfg5
    spec = importlib.util.spec_from_file_location("ref_panorama", os.path.join(REF, "RegressionNetwork", "panorama.py"))
    sys.modules.setdefault("util", MagicMock())
    pano_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pano_mod)
    P = pano_mod.Panorama(N=256, latitude=256)
    dirs, sizes, colors = raster_inputs(1, 256, 13, gutils.sphere_points(256))
    pano = P.convert_to_panorama(torch.from_numpy(dirs), torch.from_numpy(sizes),
                                 torch.from_numpy(colors)).detach().numpy()
    out["lat256/dirs"], out["lat256/sizes"], out["lat256/colors"] = dirs, sizes, colors
    out["lat256/pano_rows"] = pano[:, :, ::16]
    out["lat256/row_stride"] = np.int64(16)
    out["lat256/sum"] = np.float64(pano.astype(np.float64).sum())
    print("raster lat256", pano.shape, "sum %.3f" % pano.sum())
    np.savez_compressed(os.path.join(HERE, "rasteriser.npz"), **out)


# --------------------------------------------------------------------------- densenet
def gen_densenet():
    sys.path.insert(0, os.path.join(REF, "RegressionNetwork"))
    import DenseNet as refnet
    from oracle.densenet import deterministic_state_dict
    geomloss, gutils = ref_geomloss()
    out = {}

    def sample(t, k=64):
        flat = t.detach().reshape(-1)
        idx = np.linspace(0, flat.numel() - 1, k).astype(np.int64)
        return flat[idx].numpy(), idx

    # -- reference-native geometry: 192x256 crops, 96 anchors, B=2
    torch.manual_seed(0)
    net = refnet.DenseNet()
    net.load_state_dict(deterministic_state_dict(net.state_dict(), seed=0))
    x = torch.from_numpy(rng(0).random((2, 3, 192, 256), dtype=np.float32))
    net.eval()
    with torch.no_grad():
        pe = net(x)
        fe = net.features(x)
    for k, v in pe.items():
        out["eval/" + k] = v.numpy()
    out["eval/features_sample"], out["eval/features_idx"] = sample(fe)
    out["eval/features_mean_abs"] = np.float64(fe.abs().double().mean())

    # -- one full training step (train.py:81-102): train-mode BN, Sinkhorn blur .025
    net.train()
    crit = ref_samples_loss(geomloss, gutils, 96, 2, .025)
    g = rng(1)
    gt = {
        "distribution": torch.from_numpy(sinkhorn_inputs("sparse", 2, 96, 3)[1]),
        "intensity": torch.from_numpy(g.uniform(.05, 2, (2, 1)).astype(np.float32)),
        "rgb_ratio": torch.from_numpy(g.uniform(.4, .7, (2, 3)).astype(np.float32)),
        "ambient": torch.from_numpy(g.uniform(0, .3, (2, 3)).astype(np.float32)),
    }
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    pred = net(x)
    l2 = torch.nn.MSELoss()
    dp, dg = pred["distribution"].view(-1, 96, 1), gt["distribution"].view(-1, 96, 1)
    terms = [crit(dp, dg).sum() * 1000.0, l2(dp, dg) * 1000.0,
             l2(pred["intensity"], gt["intensity"]) * 0.1,
             l2(pred["rgb_ratio"], gt["rgb_ratio"]) * 100.0,
             l2(pred["ambient"], gt["ambient"]) * 1.0]
    loss = sum(terms)
    opt.zero_grad()
    loss.backward()
    for k, v in pred.items():
        out["train/" + k] = v.detach().numpy()
    for k, v in gt.items():
        out["train/gt_" + k] = v.numpy()
    out["train/loss_terms"] = np.array([float(t) for t in terms], dtype=np.float64)
    named = dict(net.named_parameters())
    for key in ["features.conv0.weight", "features.norm0.weight",
                "features.denseblock1.denselayer1.conv1.weight",
                "features.denseblock1.denselayer1.norm1.bias",
                "features.denseblock1.denselayer16.conv2.weight",
                "features.denseblock2.denselayer7.norm2.weight",
                "features.denseblock3.denselayer16.conv1.weight",
                "features.transition1.conv.weight", "features.transition3.norm.weight",
                "features.last_norm3.bias", "fc_dist.bias", "fc_intensity.weight"]:
        gsmp, gidx = sample(named[key].grad, 48)
        out["train/grad/" + key] = gsmp
        out["train/grad_idx/" + key] = gidx
        out["train/grad_l2/" + key] = np.float64(named[key].grad.double().norm())
    out["train/running_mean/features.norm0"] = net.features.norm0.running_mean.numpy().copy()
    out["train/running_var/features.last_norm3"] = net.features.last_norm3.running_var.numpy().copy()
    opt.step()
    out["train/post_step/fc_dist.bias"] = net.fc_dist.bias.detach().numpy().copy()
    out["train/post_step/conv0_sample"], _ = sample(net.features.conv0.weight, 48)
    print("densenet train terms", out["train/loss_terms"])

    # -- BASELINE cfg1: 1x240x320 crop -> 128 anchors, CPU forward only.  The reference
    # class raises at 240x320 (fc is 8208-wide, DenseNet.py:125); the oracle is the same
    # class with fc / fc_dist swapped for matching nn.Linear.
    torch.manual_seed(0)
    net2 = refnet.DenseNet()
    net2.fc = torch.nn.Linear(171 * 7 * 10, 1024)
    net2.fc_dist = torch.nn.Linear(1024, 128)
    net2.load_state_dict(deterministic_state_dict(net2.state_dict(), seed=1))
    net2.eval()
    x2 = torch.from_numpy(rng(2).random((1, 3, 240, 320), dtype=np.float32))
    with torch.no_grad():
        p2 = net2(x2)
    for k, v in p2.items():
        out["cfg1/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "densenet.npz"), **out)


def gen_densenet_cfg2():
    """One full training step of the reference at BASELINE's geometry (240x320 crops, 128 anchors, blur .05; cfg2's shapes
    at B = 2 so the CPU run stays in minutes) -> densenet_cfg2.npz.  The reference class raises at 240x320 (fc is
    8208-wide, DenseNet.py:125) and hard-codes 96 anchors: as for cfg1 the oracle is the SAME class with fc / fc_dist
    swapped for matching nn.Linear, everything else -- features, forward, train.py:81-102's loss -- verbatim."""
    sys.path.insert(0, os.path.join(REF, "RegressionNetwork"))
    import DenseNet as refnet
    from oracle.densenet import deterministic_state_dict
    geomloss, gutils = ref_geomloss()
    out = {}
    ln, B, crop = 128, 2, (240, 320)

    def sample(t, k=64):
        flat = t.detach().reshape(-1)
        idx = np.linspace(0, flat.numel() - 1, k).astype(np.int64)
        return flat[idx].numpy(), idx

    torch.manual_seed(0)
    net = refnet.DenseNet()
    net.fc = torch.nn.Linear(171 * 7 * 10, 1024)
    net.fc_dist = torch.nn.Linear(1024, ln)
    net.load_state_dict(deterministic_state_dict(net.state_dict(), seed=5))
    net.train()
    x = torch.from_numpy(rng(40).random((B, 3) + crop, dtype=np.float32))
    crit = ref_samples_loss(geomloss, gutils, ln, B, .05)
    g = rng(41)
    gt = {
        "distribution": torch.from_numpy(sinkhorn_inputs("sparse", B, ln, 43)[1]),
        "intensity": torch.from_numpy(g.uniform(.05, 2, (B, 1)).astype(np.float32)),
        "rgb_ratio": torch.from_numpy(g.uniform(.4, .7, (B, 3)).astype(np.float32)),
        "ambient": torch.from_numpy(g.uniform(0, .3, (B, 3)).astype(np.float32)),
    }
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    pred = net(x)
    l2 = torch.nn.MSELoss()
    dp, dg = pred["distribution"].view(-1, ln, 1), gt["distribution"].view(-1, ln, 1)
    terms = [crit(dp, dg).sum() * 1000.0, l2(dp, dg) * 1000.0,
             l2(pred["intensity"], gt["intensity"]) * 0.1,
             l2(pred["rgb_ratio"], gt["rgb_ratio"]) * 100.0,
             l2(pred["ambient"], gt["ambient"]) * 1.0]
    loss = sum(terms)
    opt.zero_grad()
    loss.backward()
    for k, v in pred.items():
        out["train/" + k] = v.detach().numpy()
    for k, v in gt.items():
        out["train/gt_" + k] = v.numpy()
    out["train/loss_terms"] = np.array([float(t) for t in terms], dtype=np.float64)
    named = dict(net.named_parameters())
    for key in ["features.conv0.weight", "features.norm0.bias",
                "features.denseblock1.denselayer1.conv1.weight",
                "features.denseblock1.denselayer2.norm1.weight",
                "features.denseblock1.denselayer15.conv1.weight",
                "features.denseblock1.denselayer16.conv2.weight",
                "features.denseblock2.denselayer8.conv1.weight",
                "features.denseblock2.denselayer7.norm2.bias",
                "features.denseblock3.denselayer16.conv1.weight",
                "features.denseblock3.denselayer1.conv2.weight",
                "features.transition1.conv.weight", "features.transition2.norm.weight",
                "features.transition3.conv.weight", "features.last_norm3.weight",
                "fc.weight", "fc_dist.bias", "fc_rgb_ratio.weight"]:
        gsmp, gidx = sample(named[key].grad, 48)
        out["train/grad/" + key] = gsmp
        out["train/grad_idx/" + key] = gidx
        out["train/grad_l2/" + key] = np.float64(named[key].grad.double().norm())
    out["train/running_mean/features.norm0"] = net.features.norm0.running_mean.numpy().copy()
    out["train/running_var/features.denseblock2.denselayer3.norm2"] = \
        net.features.denseblock2.denselayer3.norm2.running_var.numpy().copy()
    out["train/running_var/features.last_norm3"] = net.features.last_norm3.running_var.numpy().copy()
    opt.step()
    out["train/post_step/fc_dist.bias"] = net.fc_dist.bias.detach().numpy().copy()
    print("densenet cfg2-geometry train terms", out["train/loss_terms"])
    np.savez_compressed(os.path.join(HERE, "densenet_cfg2.npz"), **out)


# --------------------------------------------------------------------------- GenProjector
def projector_inputs(B, seed):
    g = rng(seed, B)
    inp = (g.random((B, 3, 128, 256)) ** 4 * 20).astype(np.float32)        # sparse-ish HDR Gaussian map
    crop = g.random((B, 3, 128, 128), dtype=np.float32)
    warped = (inp * g.uniform(0.5, 1.5, (B, 1, 128, 256))).astype(np.float32)
    mask = (g.random((B, 1, 128, 256)) > 0.7).astype(np.float32)
    return inp, crop, warped, mask


def gen_projector():
    """SPADE generator + multiscale PatchGAN of the REAL reference at ngf = ndf = 8 (the full-width G is 118 M
    parameters and 23 s per CPU forward), deterministic weights, train-mode forward + loss terms.
    VGG term excluded: pretrained VGG19 cannot be obtained offline (parity unpinned)."""
    from argparse import Namespace
    stub_io_modules()
    gp = os.path.join(REF, "GenProjector")
    sys.path.insert(0, gp)
    for m in list(sys.modules):
        if m == "util" or m.startswith("models"):
            del sys.modules[m]
    from models.networks.generator import SPADEGenerator
    from models.networks.discriminator import MultiscaleDiscriminator
    from models.networks.loss import GANLoss
    from models.pix2pix_model import Pix2PixModel
    from oracle.densenet import deterministic_projector_state_dict
    opt = Namespace(ngf=8, ndf=8, crop_size=256, aspect_ratio=2.0, num_upsampling_layers="normal",
                    norm_G="spectralspadesyncbatch3x3", norm_D="spectralinstance", norm_E="spectralinstance",
                    label_nc=3, output_nc=3, semantic_nc=3, num_D=2, n_layers_D=4, netD_subarch="n_layer",
                    no_ganFeat_loss=False, no_vgg_loss=True, gan_mode="hinge", gpu_ids=[], isTrain=True)
    torch.manual_seed(0)
    G, D = SPADEGenerator(opt), MultiscaleDiscriminator(opt)
    G.load_state_dict(deterministic_projector_state_dict(G.state_dict(), seed=11))
    D.load_state_dict(deterministic_projector_state_dict(D.state_dict(), seed=12))
    G.train(), D.train()
    inp, crop, warped, mask = (torch.from_numpy(a) for a in projector_inputs(2, 21))
    model = Pix2PixModel.__new__(Pix2PixModel)
    torch.nn.Module.__init__(model)
    model.opt, model.netG, model.netD = opt, G, D
    model.FloatTensor = torch.FloatTensor
    model.criterionGAN = GANLoss("hinge", tensor=torch.FloatTensor, opt=opt)
    model.criterionFeat = torch.nn.L1Loss()
    model.criterionVGG = lambda a, b: torch.zeros(())
    out = {}
    g_losses, fake = model.compute_generator_loss(inp, crop, warped, mask)
    total = sum(v.mean() for k, v in g_losses.items() if k != "VGG")
    total.backward()
    idx = np.linspace(0, fake.numel() - 1, 512).astype(np.int64)
    out["fake_sample"], out["fake_idx"] = fake.detach().reshape(-1)[idx].numpy(), idx
    out["fake_mean"] = np.float64(fake.detach().double().mean())
    for k in ("GAN", "GAN_Feat", "COS"):
        out["g_loss/" + k] = np.float64(g_losses[k].detach().mean())
    for key in ("sphere_conv1.weight", "head_0.conv_0.weight_orig", "up_3.norm_0.mlp_gamma.weight", "netE.fc.weight",
                "up_1.conv_s.weight_orig"):
        gsm = dict(G.named_parameters())[key].grad
        gi = np.linspace(0, gsm.numel() - 1, 32).astype(np.int64)
        out["g_grad/" + key], out["g_grad_idx/" + key] = gsm.reshape(-1)[gi].numpy(), gi
        out["g_grad_l2/" + key] = np.float64(gsm.double().norm())
    # discriminator side, with fresh identical networks (spectral-norm power iterations are stateful)
    G2, D2 = SPADEGenerator(opt), MultiscaleDiscriminator(opt)
    G2.load_state_dict(deterministic_projector_state_dict(G2.state_dict(), seed=11))
    D2.load_state_dict(deterministic_projector_state_dict(D2.state_dict(), seed=12))
    model.netG, model.netD = G2.train(), D2.train()
    d_losses = model.compute_discriminator_loss(inp, crop, warped)
    for k in ("D_Fake", "D_real"):
        out["d_loss/" + k] = np.float64(d_losses[k].detach().mean())
    with torch.no_grad():
        feats = D2(torch.cat([inp, warped], 1))
    out["d_shapes"] = np.array([[list(t.shape) for t in p] for p in feats], dtype=np.int64)
    out["d_last0"] = feats[0][-1].numpy()[:, :, ::2, ::4]
    out["d_last1"] = feats[1][-1].numpy()
    print("projector: fake mean %.4f, losses" % out["fake_mean"], {k: float(v.mean()) for k, v in g_losses.items()},
          {k: float(v) for k, v in d_losses.items()})
    np.savez_compressed(os.path.join(HERE, "projector.npz"), **out)


if __name__ == "__main__":
    install_shims()
    which = sys.argv[1:] or ["sinkhorn", "rasteriser", "densenet", "densenet_cfg2", "projector", "gmloss", "gt_param"]
    if "sinkhorn" in which:
        gen_sinkhorn()
    if "rasteriser" in which:
        gen_rasteriser()
    if "densenet" in which:
        gen_densenet()
    if "densenet_cfg2" in which:
        gen_densenet_cfg2()
    if "projector" in which:
        gen_projector()
    if "gmloss" in which:
        gen_gmloss()
    if "gt_param" in which:
        gen_gt_param()
