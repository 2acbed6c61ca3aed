// SphereConv2D of the low-resolution, wide layers (reference: GenProjector/models/networks/spherenet/sphere_cnn.py:111-124):
// the gather-GEMM with the source footprint in LDS (gather_gemm3.h) -- forward, and the input gradient as the same product
// over the transposed tap table.  These are the layers that ran as sphere_im2col + library GEMM (+ sphere_col2im) until
// round 5; with this file the 9x operand of a SphereConv exists nowhere on the training path.
#include "gather_gemm3.h"

namespace {
constexpr int kFpSmall = 256, kFpLarge = 448;
}

// 0: not supported; 1: 128 x 128 tile, 256 threads, footprint <= 256 source pixels, two workgroups per CU;
// 2: 128 x 256 tile, 512 threads, footprint <= 448 source pixels.  `fp_max`: the largest footprint (source pixels) over the
// 128-pixel tiles of a sample -- for Po < 128 the (128 / Po) whole samples a tile spans.
extern "C" int eml_sphere_conv_lowres_variant(int C, int O, int Po, int fp_max) {
  if (C < 32 || C % 32 || O < 128 || O % 128 || Po < 1 || fp_max < 1) return 0;
  if (Po % 128 != 0 && (Po > 128 || 128 % Po != 0)) return 0;
  if ((unsigned long long)O * 9 * C >= (1ull << 30) || (unsigned long long)fp_max * C * 4 >= (1ull << 32)) return 0;
  if (fp_max <= kFpSmall) return 1;
  if (fp_max <= kFpLarge && O % 256 == 0) return 2;
  return 0;
}

extern "C" int eml_sphere_conv_lowres_table_i32(const int* idx, const int* fp, int* lidx, int Po, int ke, eml_stream_t stream) {
  if (!idx || !lidx || Po < 1 || (ke != 4 && ke != 8) || ((Po % 128 == 0) != (fp != nullptr)))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_lowres_table_i32: bad arguments (Po=%d, ke=%d; fp exactly when Po %% 128 == 0)", Po, ke);
  const long n = (long)Po * 9 * ke;
  hipLaunchKernelGGL(gg3::gg3_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, fp, lidx, n, 9 * ke);
  return eml::check_launch("eml_sphere_conv_lowres_table_i32");
}

extern "C" size_t eml_sphere_conv_lowres_partial_floats(long M, int O, int split) {
  return split > 1 ? (size_t)split * (size_t)M * (size_t)O : 0;
}

extern "C" int eml_sphere_conv_lowres_f32(const float* X, const int* lidx, const float* wgt, const unsigned char* rowmax, int ke,
                                          const int* fp, int fp_max, const float* W2, const float* bias, float* Y,
                                          float* partial, int split, int B, int HW, int Po, int C, int O,
                                          const float* residual, float act_slope, eml_stream_t stream) {
  const char* what = "eml_sphere_conv_lowres_f32";
  const int* idx = lidx;   // the FOOTPRINT-LOCAL table of eml_sphere_conv_lowres_table_i32, not the raw tap table
  if (!X || !idx || !wgt || !W2 || !Y || B < 0 || HW < 1) return eml::fail(EML_EINVAL, "%s: null pointer / empty grid", what);
  const int variant = eml_sphere_conv_lowres_variant(C, O, Po, fp_max);
  if (!variant) return eml::fail(EML_EINVAL, "%s: unsupported shape (C=%d, O=%d, Po=%d, footprint %d)", what, C, O, Po, fp_max);
  if ((ke != 4 && ke != 8) || (ke == 8 && !rowmax)) return eml::fail(EML_EINVAL, "%s: ke must be 4, or 8 with rowmax", what);
  if ((Po % 128 == 0) != (fp != nullptr))
    return eml::fail(EML_EINVAL, "%s: footprint table required exactly when Po %% 128 == 0 (Po=%d)", what, Po);
  if (!fp && fp_max != (128 / Po) * HW) return eml::fail(EML_EINVAL, "%s: Po < 128 takes whole samples: fp_max must be (128 / Po) * HW", what);
  if (!(act_slope >= 0.f && act_slope <= 1.f)) return eml::fail(EML_EINVAL, "%s: act_slope %g outside [0, 1]", what, (double)act_slope);
  if (split < 1 || (C / 32) % split || (split > 1 && !partial))
    return eml::fail(EML_EINVAL, "%s: split %d must divide the %d channel chunks (and needs the partial buffer)", what, split, C / 32);
  if (B == 0) return EML_OK;
  const long M = (long)B * Po;
  if (M > 2147483647L || (unsigned long long)M * O >= (1ull << 40)) return eml::fail(EML_EINVAL, "%s: too many pixels", what);
  const int bn = variant == 1 ? 128 : 256;
  const long T = ((M + 127) / 128) * (O / bn) * split;
  const dim3 grid((unsigned)(8 * ((T + 7) / 8)));
  float* out = split > 1 ? partial : Y;
  const float* b1 = split > 1 ? nullptr : bias;
  const float* r1 = split > 1 ? nullptr : residual;
  if (variant == 1) {
    auto kern = gg3::gather_gemm3_kernel<128, 256, kFpSmall>;
    EML_ENSURE_LDS(kern, (gg3::lds_bytes<128, 256, kFpSmall>()));
    hipLaunchKernelGGL(kern, grid, dim3(256), (gg3::lds_bytes<128, 256, kFpSmall>()), (hipStream_t)stream, X, idx, wgt, W2, b1, out,
                       (int)M, HW, Po, C, O, ke, rowmax, r1, act_slope, fp, split);
  } else {
    auto kern = gg3::gather_gemm3_kernel<256, 512, kFpLarge>;
    EML_ENSURE_LDS(kern, (gg3::lds_bytes<256, 512, kFpLarge>()));
    hipLaunchKernelGGL(kern, grid, dim3(512), (gg3::lds_bytes<256, 512, kFpLarge>()), (hipStream_t)stream, X, idx, wgt, W2, b1, out,
                       (int)M, HW, Po, C, O, ke, rowmax, r1, act_slope, fp, split);
  }
  int rc = eml::check_launch(what);
  if (rc || split == 1) return rc;
  const size_t MO = (size_t)M * O;
  hipLaunchKernelGGL(gg3::gg3_reduce_kernel, dim3((unsigned)((MO / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial,
                     split, MO, O, bias, residual, act_slope, Y);
  return eml::check_launch("eml_sphere_conv_lowres_f32(reduce)");
}
