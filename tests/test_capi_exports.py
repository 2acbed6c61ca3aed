"""CPU: the C-ABI library builds, loads, and exports every symbol include/emlight_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "emlight_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eml_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    g.build()
    from emlight_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    handle = ctypes.CDLL(built_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(handle, s), "libemlight_hip.so does not export %s" % s
        assert s in built_lib.SIGNATURES, "emlight_amd/_lib.py does not bind %s" % s
    assert set(built_lib.SIGNATURES) == set(syms)


def test_loader_and_abi_version(built_lib):
    L = built_lib.lib()
    header = open(os.path.join(ROOT, "include", "emlight_hip.h")).read()
    assert L.eml_abi_version() == built_lib.ABI_VERSION == int(re.search(r"#define EML_ABI_VERSION (\d+)", header).group(1))
    assert L.eml_sinkhorn_work_floats(3, 5) == 24 * 3 * 5 + 4   # (8,B,N) planes + the 16*B*N-float exchange buffer + status


def test_argument_validation_without_gpu(built_lib):
    """Launchers validate before enqueueing: bad shapes return EML_EINVAL + a message, no GPU touched."""
    L = built_lib.lib()
    rc = L.eml_sg_rasterise_f32(None, None, None, None, 1, 4, 128, 256, None)
    assert rc == -1 and b"null" in L.eml_last_error()
    one = ctypes.c_void_p(16)
    rc = L.eml_sg_rasterise_f32(one, one, one, one, 1, 4, 128, 200, None)
    assert rc == -1 and b"W==2H" in L.eml_last_error()
    rc = L.eml_sinkhorn_fwd_f32(one, one, one, one, None, None, .05, .5, 2, -1.0, None, None, None, None, one, None, None, one, 2, 0, None)
    assert rc == -1
    rc = L.eml_sinkhorn_fwd_ex_f32(one, one, one, one, None, None, .05, .5, 2, -1.0, None, None, None, None, one, None, None, one,
                                   2, 256, 8, None)
    assert rc == -1 and b"unknown flags" in L.eml_last_error()
    # encoder / projector launchers: nulls, odd pooling sizes, misaligned channel counts
    assert L.eml_dense_pool_act_f32(one, 224, 2, 7, 8, 224, one, one, one, 224, None, None) == -1
    assert L.eml_dense_conv3x3_bwd_data_f32(one, 224, 24, one, one, one, one, one, 1, 8, 8, one, 512, one, 224, 24, None,
                                            None, None, None) == -1 and b"fused affine" in L.eml_last_error()
    assert L.eml_sphere_im2col_f32(one, one, one, None, 1, 32, 32, 8, None) == -1
    assert L.eml_sphere_col2im_f32(one, one, one, one, one, 1, 0, 32, 8, None) == -1
    assert L.eml_spade_modulate_fwd_f32(one, 16, one, 16, one, 16, 4, 16, ctypes.c_float(0.2), None) == -1   # gb needs 2C
    assert L.eml_spade_modulate_bwd_f32(one, 6, one, 6, one, 12, one, 6, one, 12, 4, 6, ctypes.c_float(1.0), None) == -1
    assert L.eml_dense_conv1x1_bwd_data_multi_f32(3, None, None, None, None, None, None, None, None, None, None, one, 224,
                                                  one, one, 10, 0, 16, one, 224, 512, None, None) == -1
    # round-2 entry points: fused SphereConv (channel counts must tile), BatchNorm fold, narrow dgrad
    assert L.eml_sphere_conv_fwd_fused_f32(one, one, one, one, None, one, 1, 32, 32, 48, 64, 4, None) == -1     # C % 32
    assert L.eml_sphere_conv_fwd_fused_f32(one, one, one, one, None, one, 1, 32, 32, 64, 96, 4, None) == -1     # O % 64
    assert b"C %" in L.eml_last_error()
    assert L.eml_sphere_conv_fwd_fused_f32(one, one, one, one, None, one, 0, 32, 32, 64, 64, 4, None) == 0      # empty batch
    assert L.eml_sphere_conv_fwd_fused_f32(one, one, one, one, None, one, 1, 32, 32, 64, 64, 2, None) == -1     # ke in {1, 4}
    assert L.eml_sphere_conv_wgrad_fused_f32(one, one, one, one, one, one, 1, 32, 32, 32, 64, 4, None) == -1  # C % 64
    assert L.eml_sphere_conv_wgrad_partial_floats(128, 256, 7) == 7 * 256 * 9 * 128
    assert L.eml_sphere_conv_dgrad_fused_f32(one, one, one, None, 8, one, one, 1, 32, 32, 64, 64, 0, None) == -1   # ke = 8 needs rowmax
    assert L.eml_sphere_conv_dgrad_fused_f32(one, one, one, None, 5, one, one, 1, 32, 32, 64, 64, 0, None) == -1   # ke in {4, 8}
    assert L.eml_bn_stats_f32(one, 6, 10, 6, one, 4, None) == -1                                               # C % 4
    assert L.eml_bn_finalize_f32(one, 8, ctypes.c_float(0.0), ctypes.c_float(0.1), one, one, None, None, None) == -1   # eps > 0
    assert L.eml_bn_bwd_apply_f32(one, 8, one, 8, 0, 8, one, one, None, one, 8, None) == 0                       # no rows
    assert L.eml_dense_conv1x1_bwd_narrow_f32(one, one, 48, 37, one, 64, one, one, one, one, 10, one, 64, one, one, 48, 4,
                                              None) == -1                                                       # odd channel offset
    # round-3 entry points
    f = ctypes.c_float
    assert L.eml_sphere_conv_fwd_fused_ex_f32(one, one, one, one, None, one, 1, 32, 32, 64, 64, 4, None, f(1.5), 0, None) == -1
    assert b"act_slope" in L.eml_last_error()
    assert L.eml_sphere_conv_fwd_fused_ex_f32(one, one, one, one, None, one, 0, 32, 32, 64, 64, 1, one, f(0.0), 0, None) == 0   # empty
    assert L.eml_sphere_conv_small_supported(3, 128) == 1 and L.eml_sphere_conv_small_supported(6, 128) == 0
    assert L.eml_sphere_conv_small_fwd_f32(one, one, one, one, None, one, 1, 32, 32, 4, 128, f(0.0), None) == -1 and b"(C, O)" in L.eml_last_error()
    assert L.eml_sphere_conv_small_fwd_f32(one, one, one, one, None, one, 0, 32, 32, 3, 128, f(0.0), None) == 0       # empty batch
    assert L.eml_sphere_conv_small_wgrad_partial_floats(32, 32768, 3, 128) == 512 * 128 * 32
    assert L.eml_sphere_conv_small_wgrad_partial_floats(1, 100, 3, 64) == 64 * 32 and L.eml_sphere_conv_small_supported(6, 64) == 0
    assert L.eml_sphere_conv_small_wgrad_f32(one, one, one, one, None, f(0.0), one, one, None, 1, 32, 32, 3, 128, None) == -1   # ReLU needs Yact
    assert L.eml_instance_norm_act_fwd_f32(one, one, one, 1, 16, 6, 1, f(1e-5), f(0.2), None) == -1    # pixel-major: C % 4
    assert L.eml_instance_norm_act_fwd_f32(one, one, one, 1, 16, 8, 1, f(1e-5), f(-0.2), None) == -1 and b"slope" in L.eml_last_error()
    assert L.eml_instance_norm_act_fwd_f32(one, one, one, 0, 16, 8, 0, f(1e-5), f(0.2), None) == 0      # empty batch
    assert L.eml_instance_norm_act_bwd_f32(one, one, one, None, 1, 16, 8, 0, f(0.2), None) == -1        # null dx
    assert L.eml_spade_norm_modulate_up2_fwd_f32(one, one, one, 1, 5, 8, 8, f(0.2), one, one, None) == -1 and b"even H" in L.eml_last_error()
    assert L.eml_spade_norm_modulate_up2_fwd_f32(one, one, one, 0, 4, 8, 8, f(0.2), one, one, None) == 0      # empty batch
    assert L.eml_spade_norm_modulate_bwd_cols_f32(one, one, one, one, one, 1, 4, 8, 6, 0, f(0.2), one, one, one, 4, None) == -1   # C % 4
    assert L.eml_spade_norm_modulate_bwd_cols_f32(one, one, one, one, one, 1, 5, 8, 8, 1, f(0.2), one, one, one, 4, None) == -1   # up2: even H
    assert L.eml_bn_bwd_apply_up2_f32(one, one, 1, 4, 8, 8, one, one, None, None, None) == -1                  # null dx
    assert L.eml_spectral_norm_scratch_floats(1024, 128) == 17 * 9 * 128 + 1024 + 2 * (5 + 1)   # t partials, t, s, 5 norm partials (f64)
    assert L.eml_spectral_norm_w2_f32(one, one, one, 1, f(0.0), one, one, one, one, 8, 4, None) == -1 and b"eps" in L.eml_last_error()
    assert L.eml_spectral_norm_w2_f32(one, one, one, 1, f(1e-12), one, one, one, one, 8, 8192, None) == -1   # row does not fit LDS
    assert L.eml_spectral_norm_w2_bwd_f32(one, one, one, one, one, None, one, 8, 4, None) == -1          # null partial
    assert L.eml_sg_rasterise_ex_f32(one, one, one, one, 1, 4, 128, 256, 6, None, None) == -1 and b"flags" in L.eml_last_error()
    assert L.eml_sg_rasterise_ex_f32(one, one, one, one, 0, 4, 128, 256, 1, None, None) == 0                    # empty batch
    assert L.eml_dense_bn_dgamma_direct_f32(one, 224, 100, 10, 10, 0, one, 48, None, 0, None, None, None, 48, one, 400,
                                            one, one, one, one, one, one, one, 64, None) == -1                  # Cin <= 384
    assert L.eml_dense_bn_dgamma_direct_f32(one, 224, 100, 9, 10, 1, one, 112, one, 112, one, one, one, 108, one, 216,
                                            one, one, one, one, one, one, one, 64, None) == -1                  # pooled: even maps
    assert L.eml_sinkhorn_fwd_f32(one, one, one, one, None, None, .05, 1.5, 2, -1.0, None, None, None, None, one, None, None,
                                  one, 2, 96, None) == -1                                                       # 0 < scaling < 1
    assert L.eml_sphere_conv_dgrad_fused_f32(one, one, one, None, 1, one, one, 0, 32, 32, 64, 64, 0, None) == 0     # ke = 1, empty batch
    assert L.eml_sphere_conv_fwd_fused_ex_f32(one, one, one, one, None, one, 1, 32, 32, 64, 64, 4, None, f(1.0), 2, None) == -1 and b"table_flags" in L.eml_last_error()
    assert L.eml_sphere_conv_small_da9_f32(one, None, f(0.0), one, one, 10, 3, 128, None) == -1                   # ReLU needs Yact
    assert L.eml_sphere_conv_small_da9_f32(one, one, f(0.0), one, one, 0, 3, 64, None) == 0                       # no rows
    # round-5 entry points: the fused L1 terms take HOST arrays of device pointers
    assert L.eml_l1_pairs_partial_doubles(3) == 3 * 1024 and L.eml_l1_pairs_partial_doubles(0) == 0
    ptrs, rows, cs, sc = (ctypes.c_void_p * 17)(*([8] * 17)), (ctypes.c_long * 17)(*([4] * 17)), (ctypes.c_int * 17)(*([4] * 17)), (ctypes.c_float * 17)()
    assert L.eml_l1_pairs_fwd_f32(0, ptrs, ptrs, ptrs, rows, cs, sc, one, one, None) == -1 and b"pairs" in L.eml_last_error()
    assert L.eml_l1_pairs_fwd_f32(17, ptrs, ptrs, ptrs, rows, cs, sc, one, one, None) == -1
    assert L.eml_l1_pairs_fwd_f32(2, ptrs, ptrs, None, rows, cs, sc, None, one, None) == -1                     # null partial
    nulls = (ctypes.c_void_p * 2)()
    assert L.eml_l1_pairs_fwd_f32(2, nulls, ptrs, None, rows, cs, sc, one, one, None) == -1 and b"pair 0" in L.eml_last_error()
    assert L.eml_l1_pairs_bwd_f32(2, ptrs, ptrs, None, rows, cs, sc, one, nulls, None, None, None) == -1       # null gradient pointer
    assert L.eml_l1_pairs_bwd_f32(2, ptrs, ptrs, None, rows, cs, sc, None, ptrs, None, None, None) == -1       # null gout


def test_product_path_has_no_cpu_fallback():
    import torch
    from emlight_amd import _lib
    from emlight_amd.RegressionNetwork.geomloss import SamplesLoss
    from emlight_amd.RegressionNetwork.util import convert_to_panorama
    x = torch.rand(2, 96, 1)
    with pytest.raises(_lib.EmlightHipError):
        SamplesLoss(anchors=96)(x, x)
    with pytest.raises(_lib.EmlightHipError):
        convert_to_panorama(torch.rand(1, 12), torch.rand(1, 4), torch.rand(1, 12))
    from emlight_amd.GenProjector.spherenet import SphereConv2D
    with pytest.raises(_lib.EmlightHipError):   # one execution path: HIP
        SphereConv2D(4, 4)(torch.rand(1, 4, 8, 16))
    from emlight_amd.RegressionNetwork.DenseNet import DenseNet
    with pytest.raises(_lib.EmlightHipError):
        DenseNet(anchors=8, crop_hw=(32, 32))(torch.rand(1, 3, 32, 32))
    with pytest.raises(TypeError):              # the stock-op engines are gone from the product
        DenseNet(engine="aten")
    with pytest.raises(NotImplementedError):    # configurations the kernels are not built for are rejected up front
        DenseNet(block_config=(24, 24, 24))
    with pytest.raises(NotImplementedError):
        DenseNet(growth_rate=32)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "emlight_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_knobs_are_validated_and_follow_the_environment(monkeypatch, capsys):
    """A/B knobs go through one parser -- a malformed value is reported and ignored (it must not break the
    import), a non-default one is announced, and a change made in-process (tests, A/B drivers) is honoured."""
    from emlight_amd import _knobs
    monkeypatch.setenv("EML_TEST_KNOB", "banana")
    assert _knobs.knob_int("EML_TEST_KNOB", 64, lo=0) == 64 and _knobs.knob_flag("EML_TEST_KNOB", True) is True
    assert "ignoring EML_TEST_KNOB" in capsys.readouterr().err
    monkeypatch.setenv("EML_TEST_KNOB", "0")
    assert _knobs.knob_int("EML_TEST_KNOB", 64, lo=0) == 0 and _knobs.knob_flag("EML_TEST_KNOB", True) is False
    assert "A/B knob EML_TEST_KNOB" in capsys.readouterr().err
    monkeypatch.setenv("EML_TEST_KNOB", "-3")
    assert _knobs.knob_int("EML_TEST_KNOB", 64, lo=0) == 64 and "ignoring" in capsys.readouterr().err   # out of range
    monkeypatch.delenv("EML_TEST_KNOB")
    assert _knobs.knob_int("EML_TEST_KNOB", 64, lo=0) == 64 and _knobs.knob_choice("EML_TEST_KNOB", "a", ("a", "b")) == "a"
    assert capsys.readouterr().err == ""


def test_guide_map_resizes_are_computed_once_per_map_and_size():
    """``networks.resized_guide`` = F.interpolate(nearest) (normalization.py:106), cached on the tensor: same object for the same
    size, the map itself at its own size, a fresh entry after an in-place change, gradients accumulate through the shared node."""
    import torch
    import torch.nn.functional as F
    from emlight_amd.GenProjector.networks import resized_guide
    x = torch.rand(2, 3, 8, 16, requires_grad=True)
    a, b = resized_guide(x, (4, 8)), resized_guide(x, (4, 8))
    assert a is b and resized_guide(x, (8, 16)) is x and torch.equal(a, F.interpolate(x, size=(4, 8), mode="nearest"))
    (a.sum() + 2 * b.sum()).backward()
    want = torch.autograd.grad(3 * F.interpolate(x, size=(4, 8), mode="nearest").sum(), x)[0]
    assert torch.equal(x.grad, want)
    y = torch.rand(1, 3, 8, 16)
    first = resized_guide(y, (2, 4))
    y.mul_(2.0)
    assert resized_guide(y, (2, 4)) is not first and torch.equal(resized_guide(y, (2, 4)), F.interpolate(y, size=(2, 4), mode="nearest"))
