"""VGG19 feature stack of the generator's perceptual loss (reference ``models/networks/architecture.py:92-125`` +
``loss.py:102-114``): ``relu1_1, relu2_1, relu3_1, relu4_1, relu5_1`` of torchvision's ``vgg19().features``.

The reference loads torchvision's ImageNet weights, which cannot be obtained offline: the WEIGHTS are
injectable -- ``VGG19Features(state_dict=...)`` takes torchvision's ``vgg19`` state dict (keys ``features.N.weight`` /
``features.N.bias``; the classifier entries are ignored) -- and when none is given the stack is initialised at random
(Kaiming, seeded) so that a training step does the reference's WORK: 23.6 GFLOP per image forward on fake and on real plus
the data gradient through fake (70.8 GFLOP per image and step) -- with a loud warning that the loss value is then not the
reference's.  Parity of the term is unpinned until real weights are supplied; the arithmetic (3x3 convolutions, ReLU, 2x2
max-pool, the five L1 terms) is pinned against the stock ops in ``tests/test_gpu_projector.py``.

Every convolution runs on the gather + f32-MFMA kernels of ``csrc/sphere_conv_fused.hip`` with the tap table of an
ordinary 3x3 convolution (``spherenet.planar_conv3x3``); activations stay channels-last.  The weights are frozen
(``requires_grad=False``, like the reference): only the input gradient of the fake branch is ever computed.
"""
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from . import spherenet

# torchvision vgg19().features up to relu5_1: index -> (in, out) of the convolutions; 'M' = max-pool 2x2
_CFG = [(0, 3, 64), (2, 64, 64), "M", (5, 64, 128), (7, 128, 128), "M", (10, 128, 256), (12, 256, 256), (14, 256, 256),
        (16, 256, 256), "M", (19, 256, 512), (21, 512, 512), (23, 512, 512), (25, 512, 512), "M", (28, 512, 512)]
_SLICE_ENDS = (1, 6, 11, 20, 29)   # the ReLU indices that close slice1..slice5 (architecture.py:103-112)


class PlanarConv3x3(nn.Module):
    """``nn.Conv2d(cin, cout, 3, padding=1)`` (same parameter names and shapes) on the HIP gather-GEMM kernels."""

    def __init__(self, cin, cout, generator=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        # torchvision's VGG initialisation; drawn from a LOCAL generator so that building the stack leaves the global CPU
        # and CUDA RNG streams (the user's seed, per-rank seeds) alone
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu", generator=generator)

    def forward(self, x, act_slope=1.0):
        return spherenet.planar_conv3x3(x, self.weight, self.bias, 1, act_slope)


class VGG19Features(nn.Module):
    """``forward(x) -> [relu1_1, relu2_1, relu3_1, relu4_1, relu5_1]`` (``architecture.py:117-124``)."""

    def __init__(self, state_dict=None, seed=0):
        super().__init__()
        gen = torch.Generator(device="cpu").manual_seed(seed)   # not torch.manual_seed: that reseeds every CUDA device too
        self.features = nn.ModuleDict()
        for item in _CFG:
            if item != "M":
                idx, cin, cout = item
                self.features[str(idx)] = PlanarConv3x3(cin, cout, generator=gen)
        self.pretrained = state_dict is not None
        self.variant = "pretrained" if self.pretrained else "random(seed=%d)" % seed
        if state_dict is not None:
            own = {k: v for k, v in state_dict.items() if k.startswith("features.") and k in self.state_dict()}
            missing = set(self.state_dict()) - set(own)
            if missing:
                raise KeyError("VGG19 state dict lacks %s" % sorted(missing))
            self.load_state_dict(own)
        else:
            warnings.warn("VGG19Features: no pretrained weights supplied (torchvision's are not obtainable offline) -- the "
                          "perceptual term runs on RANDOM features: same work as the reference's, not its loss value")
        for q in self.parameters():
            q.requires_grad = False

    def forward(self, x):
        out, idx = [], 0
        for item in _CFG:
            if item == "M":
                x = F.max_pool2d(x, kernel_size=2, stride=2)
                idx += 1
                continue
            idx = item[0]
            x = self.features[str(idx)](x, act_slope=0.0)   # conv + the ReLU that follows it (epilogue of the MFMA kernel)
            idx += 1                      # the ReLU after conv `idx`
            if idx in _SLICE_ENDS:
                out.append(x)
        return out


def vgg_loss(vgg, fake, real, weights=(1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)):
    """``VGGLoss.forward`` (``loss.py:108-114``): sum_i w_i * L1(vgg(fake)_i, vgg(real)_i.detach())."""
    xf = vgg(fake)
    with torch.no_grad():
        yf = vgg(real)
    from . import l1_terms
    if fake.is_cuda and l1_terms.ENABLED:   # the five terms in one launch each way (csrc/losses.hip)
        return l1_terms.l1_sum(zip(weights, xf, yf))
    return sum(w * F.l1_loss(a, b) for w, a, b in zip(weights, xf, yf))
