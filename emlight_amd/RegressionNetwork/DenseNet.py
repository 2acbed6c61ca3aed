"""``DenseNet`` -- EMLight's regression encoder (DenseNet-BC) on the MI355X.

Drop-in for the reference class (``RegressionNetwork/DenseNet.py:68-157``): same constructor
arguments and defaults, same ``forward(x) -> {'distribution','intensity','rgb_ratio',
'ambient'}``, and the same 625 ``state_dict`` keys
(``features.denseblock{b}.denselayer{l}.{norm1,conv1,norm2,conv2}.*``,
``features.transition{b}.{norm,conv}.*``, ``features.last_norm{b}.*``, ``fc*``), so reference
checkpoints load.  Two optional arguments lift what the reference hard-codes:
``anchors`` (reference 96, ``DenseNet.py:126``) and ``crop_hw`` (reference ``fc`` is
8208-wide == 192x256 crops, ``DenseNet.py:125``; 240x320 raises there).

There is ONE execution path: the feature extractor runs on the hand-written gfx950 kernels of
``csrc/dense_*.hip`` through ``libemlight_hip.so`` (NHWC block buffers, BN+ReLU fused into the
conv operand loads, f32 MFMA) -- see ``dense_engine.py``.  The sub-modules below are parameter
containers (they give the reference's ``state_dict`` keys); they have no stock-op ``forward``.
The same graph on stock PyTorch ops lives in ``oracle/densenet.py`` (test infrastructure).

Supported configuration of the HIP engine: ``growth_rate=12``, ``bn_size=4`` (EMLight's
constructor defaults), ``drop_rate=0``, any ``block_config`` whose widest block has at most 368
channels (3 x 16 layers: 216 / 300 / 342); backward needs train-mode BatchNorm (EMLight trains
and tests in train mode).  Anything else raises in the constructor or at the first call.
"""
import math
from collections import OrderedDict

import torch.nn as nn


class _DenseLayer(nn.Module):
    """BN1 -> ReLU -> conv1x1(4k) -> BN2 -> (no ReLU) -> conv3x3(k)  (``DenseNet.py:26-55``)."""

    def __init__(self, num_input_features, growth_rate, bn_size, drop_rate):
        super().__init__()
        inter = bn_size * growth_rate
        self.norm1 = nn.BatchNorm2d(num_input_features)
        self.conv1 = nn.Conv2d(num_input_features, inter, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(inter)
        self.conv2 = nn.Conv2d(inter, growth_rate, kernel_size=3, padding=1, bias=False)
        self.drop_rate = drop_rate

    def forward(self, x):
        raise NotImplementedError("parameter container: the layer runs inside the HIP engine (DenseNet.pooled_features)")


class _DenseBlock(nn.Module):
    def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate):
        super().__init__()
        for i in range(num_layers):
            self.add_module("denselayer%d" % (i + 1),
                            _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate))


class _Transition(nn.Module):
    """BN -> ReLU -> conv1x1 -> avgpool2  (``DenseNet.py:14-21``)."""

    def __init__(self, num_input_features, num_output_features):
        super().__init__()
        self.norm = nn.BatchNorm2d(num_input_features)
        self.conv = nn.Conv2d(num_input_features, num_output_features, kernel_size=1, stride=1, bias=False)

    def forward(self, x):
        raise NotImplementedError("parameter container: the transition runs inside the HIP engine")


class DenseNet(nn.Module):
    def __init__(self, growth_rate=12, block_config=(16, 16, 16), compression=0.5,
                 num_init_features=24, bn_size=4, drop_rate=0, avgpool_size=4,
                 anchors=96, crop_hw=(192, 256)):
        super().__init__()
        self.avgpool_size = avgpool_size
        self.growth_rate, self.bn_size = growth_rate, bn_size
        self.block_config = tuple(block_config)
        self.crop_hw = tuple(crop_hw)
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv2d(3, num_init_features, kernel_size=3, stride=1, padding=1, bias=False)),
            ("norm0", nn.BatchNorm2d(num_init_features)),
            ("relu0", nn.ReLU(inplace=True)),
        ]))
        num_features = num_init_features
        for i, num_layers in enumerate(block_config):
            self.features.add_module("denseblock%d" % (i + 1),
                                     _DenseBlock(num_layers, num_features, bn_size, growth_rate, drop_rate))
            num_features += num_layers * growth_rate
            # the reference's `i != len(block_config)` is always true (DenseNet.py:110):
            # every block is followed by a transition AND a last_norm.
            n_out = int(math.floor(num_features * compression))
            self.features.add_module("transition%d" % (i + 1), _Transition(num_features, n_out))
            num_features = n_out
            self.features.add_module("last_norm%d" % (i + 1), nn.BatchNorm2d(num_features))
        h, w = crop_hw
        for _ in block_config:
            h, w = h // 2, w // 2
        self.feat_hw = (h // avgpool_size, w // avgpool_size)
        self.fc = nn.Linear(num_features * self.feat_hw[0] * self.feat_hw[1], 1024)  # 8208 at 192x256
        self.fc_dist = nn.Linear(1024, anchors)
        self.fc_intensity = nn.Linear(1024, 1)
        self.fc_rgb_ratio = nn.Linear(1024, 3)
        self.fc_ambient = nn.Linear(1024, 3)
        if drop_rate > 0:
            raise NotImplementedError("drop_rate > 0 is not on EMLight's path (reference default 0)")
        from .dense_engine import HipDenseEncoder
        HipDenseEncoder.check_supported(self)   # reject configurations the kernels are not built for, up front
        self._hip = None

    def pooled_features(self, x):
        """relu(last_norm3(...)) average-pooled and flattened: ``(B, fc.in_features)``."""
        if self._hip is None:
            from .dense_engine import HipDenseEncoder
            self._hip = HipDenseEncoder(self)
        return self._hip(x)

    def forward(self, x):
        out = self.fc(self.pooled_features(x))
        # DenseNet.py:139-157 -- no activation between fc and the heads, none on the outputs
        return {"distribution": self.fc_dist(out),
                "intensity": self.fc_intensity(out),
                "rgb_ratio": self.fc_rgb_ratio(out),
                "ambient": self.fc_ambient(out)}
