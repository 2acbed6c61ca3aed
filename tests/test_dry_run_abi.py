"""CPU: every launcher call the host orchestration makes is checked against the bound C signature WITHOUT a GPU.

The HIP library is replaced by a recorder that validates each call's argument count and converts every argument with
the ctypes type declared in ``emlight_amd/_lib.py`` (== ``include/emlight_hip.h``, held together by
``test_capi_exports.py``).  A forward + backward of the DenseNet engine, the Sinkhorn loss and the rasteriser are then
driven on CPU tensors: the numbers are meaningless (nothing computes), but a call site that drifts from the ABI fails
here instead of on the GPU box."""
import ctypes

import pytest
import torch


class _Recorder:
    def __init__(self, signatures):
        self.signatures, self.calls, self.args = signatures, [], []
        self.returns = {}   # entry point -> value the stub returns (default 0 = EML_OK / "not supported" / 0 floats)

    def __getattr__(self, name):
        if name not in self.signatures:
            raise AttributeError(name)
        _, argtypes = self.signatures[name]

        def call(*args):
            assert len(args) == len(argtypes), "%s takes %d arguments, call site passes %d" % (name, len(argtypes), len(args))
            for k, (a, t) in enumerate(zip(args, argtypes)):
                try:
                    t.from_param(a)
                except (TypeError, ctypes.ArgumentError) as e:
                    raise AssertionError("%s: argument %d (%r) does not convert to %s" % (name, k, a, t.__name__)) from e
            self.calls.append(name)
            self.args.append((name, args))
            return self.returns.get(name, 0)
        return call


@pytest.fixture
def recorder(monkeypatch):
    from emlight_amd import _lib
    rec = _Recorder(_lib.SIGNATURES)
    monkeypatch.setattr(_lib, "lib", lambda: rec)
    monkeypatch.setattr(_lib, "current_stream", lambda: None)
    monkeypatch.setattr(_lib, "require_gpu_tensor", lambda t, name, dtype=None: t.contiguous())
    return rec


def test_densenet_engine_calls_match_the_abi(recorder):
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    from emlight_amd.RegressionNetwork.dense_engine import HipDenseEncoder
    net = DenseNet(anchors=8, crop_hw=(32, 32)).train()
    net._hip = HipDenseEncoder(net)
    net._hip._cu = 256
    out = net(torch.rand(2, 3, 32, 32))
    sum(v.sum() for v in out.values()).backward()
    for name in ("eml_dense_conv0_fwd_mfma_f32", "eml_dense_conv1x1_fwd_f32", "eml_dense_conv3x3_fwd_f32", "eml_dense_pool_act_f32",
                 "eml_dense_conv3x3_bwd_data_f32", "eml_dense_conv3x3_bwd_weight_f32", "eml_dense_conv1x1_bwd_weight_f32",
                 "eml_dense_conv1x1_bwd_data_multi_f32", "eml_dense_conv1x1_bwd_data_f32",
                 "eml_dense_bn_bwd_finalize_f32", "eml_dense_grad_materialize_f32", "eml_dense_norm0_bwd_stats_f32",
                 "eml_dense_conv0_bwd_weight_fused_f32", "eml_dense_head_pool_bwd_f32"):
        assert name in recorder.calls, name
    # the pair schedule: 8 narrow passes (riding on the upper layer's weight-gradient launch: its N12 argument, fifth
    # from the end, is set) and 8 two-layer passes per block of 16 layers
    narrow = [a for n, a in recorder.args if n == "eml_dense_conv1x1_bwd_weight_f32" and a[-3] is not None]
    assert len(narrow) == 24 and all(a[-6] % 2 == 0 for a in narrow)
    # the last pair of a block updates G; in blocks 1 and 2 (first channel a multiple of 4) every pair above it hands the next
    # pair's 24 columns over as compact tensors
    assert recorder.calls.count("eml_dense_conv1x1_bwd_data_multi_f32") == 2 + 8
    top = [a for n, a in recorder.args if n == "eml_dense_conv1x1_bwd_data_multi_top_f32"]
    assert len(top) == 14 and all(a[7] >= 48 and a[7] % 4 == 0 for a in top)    # k_hi = Cin of the pair's lower layer
    # ... which that pair then reads with a row length of 12: the upper layer's narrow operand (G, ldg) and conv3x3's (G, ldg, c0)
    assert sum(1 for a in narrow if a[-4] == 12) == 14
    # block 1's input channels: no materialize pass (blocks 2 and 3 only), its affine rides on norm0's two backward kernels
    assert recorder.calls.count("eml_dense_grad_materialize_f32") == 2 and "eml_dense_conv0_bwd_weight_f32" not in recorder.calls
    assert all(p.grad is not None for p in net.parameters())


def test_tap_packed_conv3x3_calls_match_the_abi(recorder):
    """Blocks whose width the tap-packed conv3x3 forward supports (the library says so) take that entry with its own weight
    layout (kind 3 of the batched re-layout), a band height and a grid that bn_prepare is then told about."""
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    from emlight_amd.RegressionNetwork.dense_engine import HipDenseEncoder
    recorder.returns["eml_dense_conv3x3_fwd_tp_supported"] = 4
    net = DenseNet(anchors=8, crop_hw=(32, 320)).train()
    net._hip = HipDenseEncoder(net)
    net._hip._cu = 256
    sum(v.sum() for v in net(torch.rand(2, 3, 32, 320)).values()).backward()
    tp = [a for n, a in recorder.args if n == "eml_dense_conv3x3_fwd_tp_f32"]
    assert len(tp) == 48 and "eml_dense_conv3x3_fwd_f32" not in recorder.calls   # (the stub says "4 wavefronts" for every block)
    for a in tp:
        B, H, band, grid = a[7], a[8], a[10], a[12]
        assert 4 <= band <= H and grid == B * (-(-H // band))
    # the statistics of a tap-packed layer are folded over ITS grid
    i = recorder.calls.index("eml_dense_conv3x3_fwd_tp_f32")
    nxt = recorder.args[i + 1]
    assert nxt[0] == "eml_dense_bn_prepare_f32" and nxt[1][1] == recorder.args[i][1][12]


def test_permute_table_is_rebuilt_only_when_a_pointer_moves(recorder):
    """One batched re-layout launch per pass; its device descriptor table follows the weights' data pointers."""
    import numpy as np
    from emlight_amd.RegressionNetwork.dense_engine import _PermuteTable
    tab = _PermuteTable()
    w, d = torch.zeros(48, 24), torch.zeros(32 * 48)
    tab.launch(recorder, None, [(w, d, 0, 48, 24, 32, 0)])
    first = tab.dev
    tab.launch(recorder, None, [(w, d, 0, 48, 24, 32, 0)])
    assert tab.dev is first and recorder.calls.count("eml_dense_permute_batch_f32") == 2
    host = np.frombuffer(first.numpy().tobytes(), dtype=_PermuteTable._DT)
    assert int(host["src"][0]) == w.data_ptr() and int(host["dst"][0]) == d.data_ptr() and int(host["Kp"][0]) == 32
    w2 = w.clone()                                   # e.g. model.to(): new storage -> new table
    tab.launch(recorder, None, [(w2, d, 0, 48, 24, 32, 0)])
    assert tab.dev is not first
    # an optimiser step / load_state_dict writes in place: same pointer, same table
    w2.add_(1.0)
    second = tab.dev
    tab.launch(recorder, None, [(w2, d, 0, 48, 24, 32, 0)])
    assert tab.dev is second


def test_odd_block_config_uses_the_single_layer_pass(recorder):
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    from emlight_amd.RegressionNetwork.dense_engine import HipDenseEncoder
    net = DenseNet(block_config=(2, 3, 1), anchors=8, crop_hw=(32, 32)).train()
    net._hip = HipDenseEncoder(net)
    net._hip._cu = 256
    sum(v.sum() for v in net(torch.rand(1, 3, 32, 32)).values()).backward()
    narrow = [a for n, a in recorder.args if n == "eml_dense_conv1x1_bwd_weight_f32" and a[-3] is not None]
    assert len(narrow) == 2                                                  # pairs (1,0) of block 1 and (2,1) of block 2
    assert recorder.calls.count("eml_dense_conv1x1_bwd_data_multi_f32") == 4  # + the single layers 0 of blocks 2 and 3


def test_sinkhorn_and_rasteriser_calls_match_the_abi(recorder):
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    from emlight_amd.RegressionNetwork.util import convert_to_panorama
    x = torch.rand(2, 16, 1, requires_grad=True)
    SamplesLoss(anchors=16)(x, torch.rand(2, 16, 1)).sum().backward()
    c = torch.rand(2, 48, requires_grad=True)
    convert_to_panorama(torch.rand(2, 48), torch.rand(2, 16), c, pano_hw=(8, 16)).sum().backward()
    for name in ("eml_emd_anchor_cost_f32", "eml_sinkhorn_fwd_ex_f32", "eml_sinkhorn_bwd_f32", "eml_sg_rasterise_f32",
                 "eml_sg_rasterise_bwd_colors_ex_f32"):
        assert name in recorder.calls, name


def test_spade_norm_modulate_calls_match_the_abi(recorder, monkeypatch):
    import oracle
    from emlight_amd.GenProjector import spherenet
    monkeypatch.setattr(spherenet, "_require_gpu_f32", lambda t, name: None)
    monkeypatch.setattr(spherenet, "sphere_conv", oracle.sphere_conv)   # the SphereConv itself needs real tap tables
    C, nh = 8, 4
    bn = torch.nn.BatchNorm2d(C, affine=False).train()
    g_, b_ = spherenet.SphereConv2D(nh, C), spherenet.SphereConv2D(nh, C)
    x = torch.rand(2, C, 4, 8, requires_grad=True)
    y = spherenet.spade_norm_modulate(x, bn, torch.rand(2, nh, 4, 8), g_, b_, 0.2)
    y.sum().backward()
    for name in ("eml_bn_stats_f32", "eml_bn_fold_f64", "eml_bn_finalize_f32", "eml_spade_norm_modulate_fwd_f32",
                 "eml_spade_norm_modulate_bwd_cols_f32", "eml_bn_bwd_apply_f32"):
        assert name in recorder.calls, name
    assert x.grad is not None and g_.weight.grad is not None
    bn.eval()
    n = len(recorder.calls)
    spherenet.spade_norm_modulate(x, bn, torch.rand(2, nh, 4, 8), g_, b_, 1.0).sum().backward()
    assert "eml_bn_stats_f32" not in recorder.calls[n:]   # eval: running statistics, no reduction
    # the nearest x2 upsample of the block input folded into the three passes
    bn.train()
    n = len(recorder.calls)
    xl = torch.rand(2, C, 2, 4, requires_grad=True)
    spherenet.spade_norm_modulate(xl, bn, torch.rand(2, nh, 4, 8), g_, b_, 0.2, None, up2=True).sum().backward()
    for name in ("eml_bn_stats_f32", "eml_spade_norm_modulate_up2_fwd_f32", "eml_spade_norm_modulate_bwd_cols_f32",
                 "eml_bn_bwd_apply_up2_f32"):
        assert name in recorder.calls[n:], name
    assert xl.grad.shape == xl.shape


class _FakeGeometry:
    """Tap tables of the right shapes (the real ones come from a HIP kernel): enough for the call sites' pointer arguments."""

    def __init__(self, h, w, stride):
        self.h, self.w, self.ho, self.wo = h, w, h // stride, w // stride
        n = self.ho * self.wo * 9
        self.idx, self.wgt = torch.zeros(n, 4, dtype=torch.int32), torch.zeros(n, 4)
        self.idx1 = self.wgt1 = None
        self.csr_ptr = torch.zeros(h * w + 1, dtype=torch.int32)
        self.csr_src, self.csr_w = torch.zeros(1, dtype=torch.int32), torch.zeros(1)
        self.rowshare, self.t_rowshare = 1, 0   # EML_TAP_ROWSHARE as the real geometry reports it for a stride-1 sphere table

    def transposed_table(self):
        hw = self.h * self.w
        return torch.zeros(hw * 9, 4, dtype=torch.int32), torch.zeros(hw * 9, 4), torch.zeros(hw, dtype=torch.uint8), 4


def test_one_launch_spade_and_few_output_channel_calls_match_the_abi(recorder, monkeypatch):
    """Round 4's entry points: the gamma|beta SphereConv with SPADE's modulation as its epilogue (+ its backward from gamma and
    the output), and the one-pass kernels of the layers with at most 4 output channels -- forward, weight gradient and the
    input gradient through the transposed tap table."""
    from emlight_amd.GenProjector import spherenet
    monkeypatch.setattr(spherenet, "_require_gpu_f32", lambda t, name: None)
    monkeypatch.setattr(spherenet, "sphere_geometry", lambda h, w, stride, device, kind="sphere": _FakeGeometry(h, w, stride))
    monkeypatch.setattr(spherenet, "_spade_conv_fusable", lambda x, actv, C, up2: True)
    C, nh = 64, 64
    bn = torch.nn.BatchNorm2d(C, affine=False).train()
    g_, b_ = spherenet.SphereConv2D(nh, C), spherenet.SphereConv2D(nh, C)
    for up2, hw in ((True, (2, 4)), (False, (4, 8))):
        n = len(recorder.calls)
        x, actv = torch.rand(2, C, *hw, requires_grad=True), torch.rand(2, nh, 4, 8, requires_grad=True)
        spherenet.spade_norm_modulate(x, bn, actv, g_, b_, 0.2, None, up2=up2).sum().backward()
        for name in ("eml_sphere_conv_spade_fwd_f32", "eml_spade_norm_modulate_bwd_y_f32",
                     "eml_bn_bwd_apply_up2_f32" if up2 else "eml_bn_bwd_apply_f32"):
            assert name in recorder.calls[n:], name
        assert "eml_spade_norm_modulate_bwd_cols_f32" not in recorder.calls[n:]
        assert x.grad.shape == x.shape and actv.grad is not None and g_.weight.grad is not None and b_.bias.grad is not None
    with pytest.raises(ValueError, match="guide map"):
        spherenet.spade_norm_modulate(torch.rand(2, C, 3, 8), bn, torch.rand(2, nh, 4, 8), g_, b_, 0.2, None, up2=False)
    with torch.no_grad():   # the discriminator step's generator pass: gamma is not kept (NULL)
        spherenet.spade_norm_modulate(torch.rand(2, C, 4, 8), bn, torch.rand(2, nh, 4, 8), g_, b_, 0.2, None)
    assert [a for nme, a in recorder.args if nme == "eml_sphere_conv_spade_fwd_f32"][-1][9] is None
    # conv_img 64 -> 3
    recorder.returns["eml_sphere_conv_narrow_supported"] = 1
    conv = spherenet.SphereConv2D(64, 3)
    n = len(recorder.calls)
    xi = torch.rand(2, 64, 4, 8, requires_grad=True)
    conv(xi).sum().backward()
    for name in ("eml_sphere_conv_narrow_scratch_floats", "eml_sphere_conv_narrow_fwd2_f32", "eml_sphere_conv_narrow_wgrad2_partial_floats",
                 "eml_sphere_conv_narrow_wgrad2_f32", "eml_sphere_conv_narrow_dgrad2_f32"):   # round 6: project, then gather
        assert name in recorder.calls[n:], name
    assert "eml_sphere_im2col_f32" not in recorder.calls[n:] and "eml_sphere_col2im_f32" not in recorder.calls[n:]
    assert xi.grad.shape == xi.shape and conv.weight.grad.shape == conv.weight.shape
    # the input gradient reuses the V the weight gradient left in the scratch tensor (scratch_has_v = 1)
    dg = [a for nme, a in recorder.args[n:] if nme == "eml_sphere_conv_narrow_dgrad2_f32"][-1]
    assert dg[8] == 1
    # EML_NARROW_PROJECT=0: the one-pass kernels
    monkeypatch.setattr(spherenet.SphereConv2D, "narrow_project", False)
    n = len(recorder.calls)
    conv(torch.rand(2, 64, 4, 8, requires_grad=True)).sum().backward()
    for name in ("eml_sphere_conv_narrow_fwd_f32", "eml_sphere_conv_narrow_wgrad_partial_floats", "eml_sphere_conv_narrow_wgrad_f32",
                 "eml_sphere_conv_narrow_dgrad_f32"):
        assert name in recorder.calls[n:], name


def test_batched_spectral_norm_calls_match_the_abi(recorder, monkeypatch):
    """Round 5: ``spectral_precompute`` normalises every fused-hook weight under a root in one batch call; each hook then
    picks its result up once, and falls back to its own launches when the weight changed in between."""
    from emlight_amd.GenProjector import spherenet
    hook_t = spherenet._FusedSpectralNormHook
    monkeypatch.setattr(hook_t, "eligible", staticmethod(lambda sn, w: w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)))
    root = torch.nn.Sequential(spherenet.fused_spectral_norm(spherenet.SphereConv2D(8, 16)),
                               spherenet.fused_spectral_norm(spherenet.SphereConv2D(16, 4)),
                               spherenet.fused_spectral_norm(torch.nn.Conv2d(4, 4, 3)))
    hooks = [h for m in root for h in m._forward_pre_hooks.values()]
    assert all(isinstance(h, hook_t) for h in hooks) and len(hooks) == 3
    spherenet.spectral_precompute(root)
    assert recorder.calls.count("eml_spectral_norm_w2_batch_f32") == 1
    name, args = recorder.args[-1]
    assert args[0] == 3 and list(args[10]) == [16, 4, 4] and list(args[11]) == [8, 16, 4] and args[4] == 1   # n, O[], C[], iterate
    assert all(h.pre is not None for h in hooks)
    n = len(recorder.calls)
    hooks[0](root[0], ())
    assert hooks[0].pre is None and "eml_spectral_norm_w2_f32" not in recorder.calls[n:]     # picked up, once
    assert tuple(root[0].weight.shape) == (16, 8, 3, 3)
    hooks[0](root[0], ())
    assert recorder.calls[n:].count("eml_spectral_norm_w2_f32") == 1                         # a second call iterates itself
    with torch.no_grad():
        root[1].weight_orig.mul_(2.0)                                                        # e.g. an optimizer step in between
    hooks[1](root[1], ())
    assert recorder.calls[n:].count("eml_spectral_norm_w2_f32") == 2 and hooks[1].pre is None
    root.eval()
    spherenet.spectral_precompute(root)
    assert recorder.args[-1][1][4] == 0                                                      # eval: stored u, v
    monkeypatch.setattr(spherenet, "_sn_batch", False)
    n = len(recorder.calls)
    spherenet.spectral_precompute(root)
    assert len(recorder.calls) == n
