"""ctypes binding of libemlight_hip.so (the C ABI declared in include/emlight_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
``lib()`` raises -- a GPU box must run the HIP kernels or fail loudly.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EML_LIB_PATH") or os.path.join(_HERE, "libemlight_hip.so")   # override: A/B builds (tools/exp_build.sh)

_f32p = ctypes.c_void_p   # device pointers travel as integers (tensor.data_ptr())
_i32p = ctypes.c_void_p
_f64p = ctypes.c_void_p
_stream = ctypes.c_void_p
_int = ctypes.c_int

ABI_VERSION = 29   # == EML_ABI_VERSION of include/emlight_hip.h

# symbol -> (restype, argtypes): exactly the declarations of include/emlight_hip.h
SIGNATURES = {
    "eml_abi_version": (_int, []),
    "eml_last_error": (ctypes.c_char_p, []),
    "eml_sg_rasterise_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _stream]),
    "eml_sg_rasterise_ex_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, ctypes.c_void_p, _stream]),
    "eml_sg_rasterise_bwd_colors_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _stream]),
    "eml_emd_anchor_cost_f32": (_int, [_f32p, _f32p, _int, _stream]),
    "eml_sinkhorn_schedule_f32": (_int, [_f32p, _f32p, ctypes.c_long, ctypes.c_double, ctypes.c_double, _int,
                                         ctypes.c_double, _f32p, _f32p, _i32p, _f32p, _stream]),
    "eml_sinkhorn_work_floats": (ctypes.c_size_t, [_int, _int]),
    "eml_sinkhorn_fwd_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_double, ctypes.c_double, _int,
                                    ctypes.c_double, _f32p, _f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int,
                                    _stream]),
    "eml_sinkhorn_fwd_ex_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_double, ctypes.c_double, _int,
                                       ctypes.c_double, _f32p, _f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int,
                                       _int, _stream]),
    "eml_sinkhorn_bwd_f32": (_int, [_f32p, _f32p, _f32p, _int, _int, _stream]),
    # the generator's L1-type loss terms (host pointer arrays: ctypes arrays of c_void_p / c_long / c_int / c_float)
    "eml_l1_pairs_partial_doubles": (ctypes.c_size_t, [_int]),
    "eml_l1_pairs_fwd_f32": (_int, [_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, _f64p, _f32p, _stream]),
    "eml_l1_pairs_bwd_f32": (_int, [_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, _f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _stream]),
    "eml_colsum_partial_doubles": (ctypes.c_size_t, [_int]),
    "eml_colsum_f32": (_int, [_f32p, ctypes.c_long, _int, _f64p, _f32p, _stream]),
    # ground-truth parametrisation
    "eml_gt_anchor_index_i32": (_int, [_f32p, _int, _int, _int, _i32p, _stream]),
    "eml_gt_parametrise_f64": (_int, [_f32p, _i32p, _i32p, _int, _int, _int, _int, _f32p, _f32p, _f32p, _stream]),
    # GenProjector SphereConv2D
    "eml_sphere_tap_table_f32": (_int, [_f32p, _int, _int, _int, _int, _i32p, _f32p, _stream]),
    "eml_sphere_im2col_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _int, _int, _int, _int, _stream]),
    "eml_sphere_col2im_f32": (_int, [_f32p, _i32p, _i32p, _f32p, _f32p, _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_fwd_fused_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _int,
                                             _stream]),
    "eml_sphere_conv_fwd_fused_ex_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _int,
                                                _f32p, ctypes.c_float, _int, _stream]),
    "eml_sphere_conv_small_supported": (_int, [_int, _int]),
    "eml_sphere_conv_small_fwd_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int,
                                             ctypes.c_float, _stream]),
    "eml_sphere_conv_small_wgrad_partial_floats": (ctypes.c_size_t, [_int, _int, _int, _int]),
    "eml_sphere_conv_small_wgrad_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, ctypes.c_float, _f32p, _f32p, _f32p, _int,
                                               _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_small_da9_f32": (_int, [_f32p, _f32p, ctypes.c_float, _f32p, _f32p, ctypes.c_long, _int, _int, _stream]),
    "eml_spade_norm_modulate_up2_fwd_f32": (_int, [_f32p, _f32p, _f32p, _int, _int, _int, _int, ctypes.c_float, _f32p, _f32p,
                                                   _stream]),
    "eml_spade_norm_modulate_bwd_cols_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int,
                                                    ctypes.c_float, _f32p, _f32p, _f32p, _int, _stream]),
    "eml_sg_rasterise_bwd_work_floats": (ctypes.c_size_t, [_int, _int, _int, _int]),
    "eml_sg_rasterise_bwd_colors_ex_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_narrow_supported": (_int, [_int, _int]),
    "eml_sphere_conv_narrow_fwd_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_narrow_dgrad_f32": (_int, [_f32p, _i32p, _f32p, _int, _f32p, _f32p, _int, _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_narrow_wgrad_partial_floats": (ctypes.c_size_t, [_int, _int, _int, _int]),
    "eml_sphere_conv_narrow_wgrad_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_narrow_scratch_floats": (ctypes.c_size_t, [_int, _int]),
    "eml_sphere_conv_narrow_fwd2_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _stream]),
    "eml_sphere_conv_narrow_wgrad2_partial_floats": (ctypes.c_size_t, [_int, _int, _int]),
    "eml_sphere_conv_narrow_dgrad2_f32": (_int, [_f32p, _i32p, _f32p, _int, ctypes.c_void_p, _f32p, _f32p, _f32p, _int, _int, _int,
                                                 _int, _int, _int, _stream]),
    "eml_sphere_conv_narrow_wgrad2_f32": (_int, [_f32p, _i32p, _f32p, _int, ctypes.c_void_p, _f32p, _f32p, _f32p, _f32p, _int, _int,
                                                 _int, _int, _int, _stream]),
    "eml_sphere_conv_spade_supported": (_int, [_int, _int, ctypes.c_long]),
    "eml_sphere_conv_spade_fwd_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int,
                                             _int, _int, _int, _int, ctypes.c_float, _int, _stream]),
    "eml_spade_norm_modulate_bwd_y_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int,
                                                 ctypes.c_float, _f32p, _f32p, _f32p, _int, _stream]),
    "eml_bn_bwd_apply_up2_f32": (_int, [_f32p, _f32p, _int, _int, _int, _int, _f32p, _f32p, _f32p, _f32p, _stream]),
    "eml_spectral_norm_scratch_floats": (ctypes.c_size_t, [_int, _int]),
    "eml_spectral_norm_w2_f32": (_int, [_f32p, _f32p, _f32p, _int, ctypes.c_float, _f32p, _f32p, _f32p, _f32p, _int, _int,
                                        _stream]),
    "eml_spectral_norm_w2_bwd_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _stream]),
    "eml_spectral_norm_w2_batch_f32": (_int, [_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _int, ctypes.c_float,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, _stream]),
    "eml_spade_heads_w2_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _stream]),
    "eml_instance_norm_act_fwd_f32": (_int, [_f32p, _f32p, _f32p, _int, _int, _int, _int, ctypes.c_float, ctypes.c_float,
                                             _stream]),
    "eml_instance_norm_act_bwd_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, ctypes.c_float, _stream]),
    "eml_sphere_conv_dgrad_fused_f32": (_int, [_f32p, _i32p, _f32p, _i32p, _int, _f32p, _f32p, _int, _int, _int, _int, _int,
                                               _int, _stream]),
    "eml_sphere_conv_lowres_variant": (_int, [_int, _int, _int, _int]),
    "eml_sphere_conv_lowres_table_i32": (_int, [_i32p, _i32p, _i32p, _int, _int, _stream]),
    "eml_sphere_conv_lowres_partial_floats": (ctypes.c_size_t, [ctypes.c_long, _int, _int]),
    "eml_sphere_conv_lowres_f32": (_int, [_f32p, _i32p, _f32p, _i32p, _int, _i32p, _int, _f32p, _f32p, _f32p, _f32p, _int, _int,
                                          _int, _int, _int, _int, _f32p, ctypes.c_float, _stream]),
    "eml_sphere_conv_wgrad_partial_floats": (ctypes.c_size_t, [_int, _int, _int]),
    "eml_sphere_conv_wgrad_fused_f32": (_int, [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _int,
                                               _stream]),
    "eml_spade_modulate_fwd_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _int, ctypes.c_long, _int, ctypes.c_float,
                                          _stream]),
    "eml_spade_modulate_bwd_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _int, _f32p, _int, _f32p, _int, ctypes.c_long,
                                          _int, ctypes.c_float, _stream]),
    "eml_bn_stats_f32": (_int, [_f32p, _int, ctypes.c_long, _int, _f32p, _int, _stream]),
    "eml_bn_fold_f64": (_int, [_f32p, _int, _int, _f32p, _stream]),
    "eml_bn_finalize_f32": (_int, [_f32p, _int, ctypes.c_float, ctypes.c_float, _f32p, _f32p, _f32p, _f32p, _stream]),
    "eml_spade_norm_modulate_fwd_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _int, ctypes.c_long, _int, ctypes.c_float,
                                               _f32p, _f32p, _stream]),
    "eml_bn_bwd_apply_f32": (_int, [_f32p, _int, _f32p, _int, ctypes.c_long, _int, _f32p, _f32p, _f32p, _f32p, _int,
                                    _stream]),
    # DenseNet-BC encoder, forward
    "eml_dense_conv0_fwd_f32": (_int, [_f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _f32p, _int, _stream]),
    "eml_dense_conv0_fwd_mfma_f32": (_int, [_f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _f32p, _int, _stream]),
    "eml_dense_bn_apply_f32": (_int, [_f32p, _int, _f32p, _int, _int, ctypes.c_long, _f32p, _f32p, _int, _f32p,
                                      _int, _stream]),
    "eml_dense_bn_prepare_f32": (_int, [_f32p, _int, _int, _int, _int, ctypes.c_double, _f32p, _f32p, _f32p, _f32p,
                                        _f32p, _f32p, _f32p, _int, _int, ctypes.c_float, ctypes.c_float, _int,
                                        _f32p, _f32p, _stream]),
    "eml_dense_permute_w1_f32": (_int, [_f32p, _int, _int, _int, _f32p, _stream]),
    "eml_dense_permute_w2_f32": (_int, [_f32p, _int, _f32p, _stream]),
    "eml_dense_permute_batch_f32": (_int, [ctypes.c_void_p, _int, _stream]),
    "eml_dense_conv1x1_fwd_f32": (_int, [_f32p, _int, ctypes.c_long, _int, _int, _int, _int, _f32p, _f32p, _f32p,
                                         _int, _f32p, _int, _f32p, _int, ctypes.c_void_p, _stream]),
    "eml_dense_conv3x3_fwd_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _f32p,
                                         _int, _stream]),
    "eml_dense_permute_w2_tp_f32": (_int, [_f32p, _f32p, _stream]),
    "eml_dense_conv3x3_fwd_tp_supported": (_int, [_int, _int, _int]),
    "eml_dense_conv3x3_fwd_tp_f32": (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _int, _f32p,
                                            _int, _stream]),
    "eml_dense_pool_act_f32": (_int, [_f32p, _int, _int, _int, _int, _int, _f32p, _f32p, _f32p, _int, ctypes.c_void_p,
                                      _stream]),
    "eml_dense_head_pool_fwd_f32": (_int, [_f32p, _int, _int, _int, _int, _int, _int, _f32p, _stream]),
    # DenseNet-BC encoder, backward
    "eml_dense_conv3x3_bwd_data_f32": (_int, [_f32p, _int, _int, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int,
                                              _f32p, _int, _f32p, _int, _int, _f32p, _f32p, _f32p, _stream]),
    "eml_dense_conv3x3_bwd_fused_supported": (_int, [_int, _int, _int, _int]),
    "eml_dense_conv3x3_bwd_fused_f32": (_int, [_f32p, _int, _int, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _f64p, _int,
                                               _f32p, _int, _int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _stream]),
    "eml_dense_conv3x3_bwd_weight_f32": (_int, [_f32p, _int, _int, _f32p, _f32p, _f32p, _int, _int, _int, _f32p,
                                                _f32p, _int, _stream]),
    "eml_dense_bn_bwd_finalize_f32": (_int, [_f32p, _int, _int, ctypes.c_double, _f32p, _f32p, _f32p, _int, _int,
                                             _int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int,
                                             _f32p, _f32p, _f32p, _int, _i32p, _i32p, _stream]),
    "eml_dense_bn_dgamma_direct_f32": (_int, [_f32p, _int, ctypes.c_long, _int, _int, _int, _f32p, _int, _f32p, _int,
                                              _f32p, _f32p, _f32p, _int, _f32p, _int, _f32p, _f32p, _f32p, _f32p, _i32p,
                                              _f32p, _f32p, _int, _stream]),
    "eml_dense_conv1x1_bwd_weight_f32": (_int, [_f32p, _int, ctypes.c_long, _int, _int, _int, _int, _int, _f32p,
                                                _f32p, _f32p, _int, _f32p, _int, _f32p, _f32p, _f32p, _int, _f32p,
                                                _f32p, _int, _f32p, _f32p, _int, _f32p, _int, _f32p, _f32p, _stream]),
    "eml_dense_conv1x1_bwd_narrow_f32": (_int, [_f32p, _f32p, _int, _int, _f32p, _int, _f32p, _f32p, _f32p, _f32p,
                                                ctypes.c_long, _f32p, _int, _f32p, _f32p, _int, _int, _stream]),
    "eml_dense_permute_w1_bwd_f32": (_int, [_f32p, _int, _int, _int, _int, _f32p, _stream]),
    "eml_dense_conv1x1_bwd_data_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _f32p, _f32p, _int, _f32p, _f32p,
                                              _int, _f32p, _f32p, _f32p, _f32p, ctypes.c_long, _int, _int, _int,
                                              _int, _f32p, _int, _int, _f32p, _int, ctypes.c_void_p, _stream]),
    "eml_dense_conv1x1_bwd_data_multi_f32": (_int, [_int] + [ctypes.c_void_p] * 10 + [_f32p, _int, _f32p, _f32p,
                                                    ctypes.c_long, _int, _int, _f32p, _int, _int, ctypes.c_void_p,
                                                    _stream]),
    "eml_dense_conv1x1_bwd_data_multi_top_f32": (_int, [ctypes.c_void_p] * 6 + [ctypes.c_long, _int, _f32p, _int, _int,
                                                        ctypes.c_void_p, _f32p, _stream]),
    "eml_dense_grad_materialize_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _f32p, _int, _int, ctypes.c_long,
                                              _stream]),
    "eml_dense_bn_bwd_stats_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _int, _int, _int, ctypes.c_long, _f32p,
                                          _f32p, _f32p, _int, _stream]),
    "eml_dense_norm0_bwd_stats_f32": (_int, [_f32p, _int, _f32p, _int, _f32p, _f32p, _f32p, _f32p, _int, ctypes.c_long, _f32p, _f32p,
                                             ctypes.c_void_p, _int, _stream]),
    "eml_dense_conv0_bwd_weight_fused_f32": (_int, [_f32p, _f32p, _int, _f32p, _int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                                    _int, _int, _int, _f32p, _f32p, _int, _stream]),
    "eml_dense_conv0_bwd_weight_f32": (_int, [_f32p, _f32p, _int, _f32p, _int, _f32p, _int, _f32p, _f32p, _f32p,
                                              _int, _int, _int, _f32p, _f32p, _int, _stream]),
    "eml_dense_head_pool_bwd_f32": (_int, [_f32p, _f32p, _int, _int, _int, _int, _int, _int, _f32p, _int, _stream]),
}

_lock = threading.Lock()
_lib = None


class EmlightHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EmlightHipError(
                "libemlight_hip.so not found at %s -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C emlight_amd/csrc`. "
                "There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise EmlightHipError("libemlight_hip.so lacks symbol %s" % name) from e
            fn.restype, fn.argtypes = res, args
        got = handle.eml_abi_version()
        if got != ABI_VERSION:  # a stale in-tree .so bound with the wrong argument lists would corrupt memory, not fail
            raise EmlightHipError("libemlight_hip.so has ABI version %d, this binding expects %d -- rebuild it "
                                  "(make -C emlight_amd/csrc)" % (got, ABI_VERSION))
        _lib = handle
    # Loading the library changes nothing else in the process unless an ENTRY POINT asked for it
    # (_runtime.entry_point_defaults(): bench.py, the train / test / joint mains, the test session): the recorded library-GEMM
    # selection -- TunableOp in look-up mode, process-wide -- is then switched on here, once the rank's device is chosen
    from . import _gemm_selection
    _gemm_selection.ensure_if_requested()
    return _lib


_DEBUG_SYNC = bool(os.environ.get("EML_DEBUG_SYNC"))


def check(rc, what):
    if rc != 0:
        msg = lib().eml_last_error()
        raise EmlightHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
    if _DEBUG_SYNC:  # debugging aid: localise an asynchronous GPU fault to the launcher that caused it
        import sys
        import torch
        sys.stderr.write("[eml] %s ... " % what)
        sys.stderr.flush()
        torch.cuda.synchronize()
        sys.stderr.write("ok\n")


def ptr(t):
    """Device pointer of a contiguous f32/i32 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu_tensor(t, name, dtype=None):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise EmlightHipError("%s must be a tensor on the MI355X (got %s); there is no CPU path"
                              % (name, getattr(t, "device", type(t))))
    if t.dtype != (dtype or torch.float32):
        raise EmlightHipError("%s must be %s (got %s)" % (name, dtype or torch.float32, t.dtype))
    return t.contiguous()
