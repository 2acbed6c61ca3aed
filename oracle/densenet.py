"""Oracle (test infrastructure): EMLight's regression encoder and its training loss.

CPU f32 restatement of ``RegressionNetwork/DenseNet.py`` (DenseNet-BC, k=12, blocks
(16,16,16), 24 stem features, compression .5, no stem pooling -- ``DenseNet.py:82-93``)
as plain ``torch.nn.functional`` calls over a module tree whose ``state_dict`` keys are
exactly the reference's (``features.denseblock{b}.denselayer{l}.{norm1,conv1,norm2,conv2}``,
``features.transition{b}.{norm,conv}``, ``features.last_norm{b}``, ``fc``, ``fc_*``).

Parameterised where the reference hard-codes: ``anchors`` (reference 96,
``DenseNet.py:126``) and ``crop_hw`` (reference fc.in_features 8208 == 192x256,
``DenseNet.py:125``).
"""
import math
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Norm(nn.BatchNorm2d):
    pass


def _mk_layer(cin, growth, bn_size):
    m = nn.Module()
    m.norm1 = _Norm(cin)
    m.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
    m.norm2 = _Norm(bn_size * growth)
    m.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)
    return m


def _mk_transition(cin, cout):
    m = nn.Module()
    m.norm = _Norm(cin)
    m.conv = nn.Conv2d(cin, cout, 1, bias=False)
    return m


def _bn(x, bn, training):
    if training:
        return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                            True, bn.momentum, bn.eps)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                        False, 0.0, bn.eps)


class OracleDenseNet(nn.Module):
    """Restates ``DenseNet.DenseNet`` (``DenseNet.py:68-157``)."""

    def __init__(self, growth_rate=12, block_config=(16, 16, 16), compression=0.5,
                 num_init_features=24, bn_size=4, avgpool_size=4, anchors=96,
                 crop_hw=(192, 256)):
        super().__init__()
        self.block_config = tuple(block_config)
        self.avgpool_size = avgpool_size
        # tests at BASELINE's full batch (64 x 240 x 320) set this: every dense layer is re-run in the backward instead of
        # keeping its intermediates (same arithmetic; 4x less memory, which is what lets the f64 yardstick fit next to it)
        self.checkpoint_layers = False
        f = nn.Module()
        f.conv0 = nn.Conv2d(3, num_init_features, 3, padding=1, bias=False)
        f.norm0 = _Norm(num_init_features)
        c = num_init_features
        for b, n_layers in enumerate(block_config, 1):
            blk = nn.Module()
            for l in range(n_layers):
                setattr(blk, "denselayer%d" % (l + 1), _mk_layer(c + l * growth_rate, growth_rate, bn_size))
            setattr(f, "denseblock%d" % b, blk)
            c += n_layers * growth_rate
            # DenseNet.py:110 -- the `i != len(block_config)` test is always true:
            # a transition + last_norm follows EVERY block.
            cout = int(math.floor(c * compression))
            setattr(f, "transition%d" % b, _mk_transition(c, cout))
            c = cout
            setattr(f, "last_norm%d" % b, _Norm(c))
        self.features = f
        h, w = crop_hw
        for _ in block_config:
            h, w = h // 2, w // 2
        self.fc = nn.Linear(c * (h // avgpool_size) * (w // avgpool_size), 1024)
        self.fc_dist = nn.Linear(1024, anchors)
        self.fc_intensity = nn.Linear(1024, 1)
        self.fc_rgb_ratio = nn.Linear(1024, 3)
        self.fc_ambient = nn.Linear(1024, 3)

    def features_forward(self, x):
        f, tr = self.features, self.training
        x = F.relu(_bn(F.conv2d(x, f.conv0.weight, padding=1), f.norm0, tr))
        for b, n_layers in enumerate(self.block_config, 1):
            blk = getattr(f, "denseblock%d" % b)
            for l in range(n_layers):
                L = getattr(blk, "denselayer%d" % (l + 1))

                def layer(x, L=L):
                    # DenseNet.py:30-43: BN1 -> ReLU -> 1x1 -> BN2 -> (no ReLU) -> 3x3 -> cat
                    z = F.conv2d(F.relu(_bn(x, L.norm1, tr)), L.conv1.weight)
                    return F.conv2d(_bn(z, L.norm2, tr), L.conv2.weight, padding=1)
                if self.checkpoint_layers and torch.is_grad_enabled():
                    from torch.utils.checkpoint import checkpoint
                    new = checkpoint(layer, x, use_reentrant=False)
                else:
                    new = layer(x)
                x = torch.cat([x, new], 1)
            T = getattr(f, "transition%d" % b)
            # DenseNet.py:14-21: BN -> ReLU -> 1x1 -> avgpool2
            x = F.avg_pool2d(F.conv2d(F.relu(_bn(x, T.norm, tr)), T.conv.weight), 2, 2)
            x = _bn(x, getattr(f, "last_norm%d" % b), tr)
        return x

    def forward(self, x):
        feat = self.features_forward(x)
        # DenseNet.py:135-157: relu -> avgpool(4) -> flatten -> fc -> four linear heads,
        # no activation between fc and the heads, no output activation.
        out = F.avg_pool2d(F.relu(feat), self.avgpool_size).reshape(feat.size(0), -1)
        out = self.fc(out)
        return {"distribution": self.fc_dist(out), "intensity": self.fc_intensity(out),
                "rgb_ratio": self.fc_rgb_ratio(out), "ambient": self.fc_ambient(out)}


def deterministic_state_dict(reference_state_dict, seed=0):
    """Key-addressed deterministic weights (independent of module construction order).

    Every tensor is drawn from a numpy Generator seeded by ``seed`` and the CRC32 of its
    state-dict key, so the reference module (golden generation), the oracle and the HIP
    model all load bit-identical parameters without shipping a 37 MB checkpoint.
    """
    out = {}
    for key, ref in reference_state_dict.items():
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.long)
            continue
        if key.endswith("running_mean"):
            v = rng.uniform(-0.1, 0.1, shape)
        elif key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif "norm" in key and key.endswith("weight"):
            v = rng.uniform(0.5, 1.5, shape)
        elif "norm" in key and key.endswith("bias"):
            v = rng.uniform(-0.2, 0.2, shape)
        elif key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, math.sqrt(2.0 / fan_in), shape)
        else:  # linear bias
            v = rng.uniform(-0.05, 0.05, shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return out


def regression_loss(pred, gt, emd_fn, anchors):
    """Weighted training loss of ``RegressionNetwork/train.py:90-98``.

    1000*sum(EMD) + 1000*MSE(dist) + 0.1*MSE(intensity) + 100*MSE(rgb) + 1*MSE(ambient).
    """
    dp = pred["distribution"].view(-1, anchors, 1)
    dg = gt["distribution"].view(-1, anchors, 1)
    terms = {
        "dist_emloss": emd_fn(dp, dg).sum() * 1000.0,
        "dist_l2loss": F.mse_loss(dp, dg) * 1000.0,
        "intensity_loss": F.mse_loss(pred["intensity"], gt["intensity"]) * 0.1,
        "rgb_loss": F.mse_loss(pred["rgb_ratio"], gt["rgb_ratio"]) * 100.0,
        "ambient_loss": F.mse_loss(pred["ambient"], gt["ambient"]) * 1.0,
    }
    return sum(terms.values()), terms


def deterministic_projector_state_dict(reference_state_dict, seed=0):
    """Key-addressed deterministic weights for the GenProjector networks (spectral-norm aware):
    conv / linear weights ~ N(0, 1/fan_in), power-iteration vectors unit-norm, small biases."""
    out = {}
    for key, ref in reference_state_dict.items():
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.long)
            continue
        if key.endswith("running_mean"):
            v = rng.uniform(-0.1, 0.1, shape)
        elif key.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("weight_u") or key.endswith("weight_v"):
            v = rng.standard_normal(shape)
            v = v / np.linalg.norm(v)
        elif key.endswith("weight") or key.endswith("weight_orig"):
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            v = rng.normal(0.0, math.sqrt(1.0 / fan_in), shape)
        else:
            v = rng.uniform(-0.05, 0.05, shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return out
