// SphereConv2D for the few-channel OUTPUT layers of the GenProjector: the generator's conv_img 64 -> 3 at full resolution
// (generator.py:60, 84-86) and the discriminators' final 512 -> 1..3 convolutions (discriminator.py:70-74).
//
// With O <= 4 output channels the product is not a GEMM worth a matrix unit: per pixel it reads 9 * C interpolated values
// for 2 * 9 * C * O flops.  On the general path these layers were the largest users of the 9x im2col operand left in the
// step -- conv_img at B = 32, 128 x 256: a 2.4 GB A9 written, read by an N = 3 library GEMM, and in the backward a 2.4 GB
// dA9 written by another skinny GEMM and read by col2im (profiles/r04_im2col_audit.txt).  Here nothing 9x-sized exists:
//   forward : 16 lanes x float4 = 64 channels of one output pixel; the 9 taps are gathered (4 bilinear corners each,
//             16-byte loads, L1/L2 resident: the input is read from HBM once) and multiplied by the lane's slice of W2
//             (LDS, (tap, c/4, o) order: O consecutive float4 per lane), then a 16-lane butterfly sum; lane 0 writes O floats.
//   dgrad   : per INPUT pixel q, lanes 0..8 of its 16-lane group reduce "their" tap's entries of the transposed tap table
//             to v[tap][o] = sum_s w_s dY[p_s][o] (27 numbers), a 16-wide shuffle hands them to every lane, and
//             dX[q][c] = sum_{tap,o} v[tap][o] W2[o][tap*C + c] -- a gather, deterministic, no atomics.
//   wgrad   : dW2[o][tap*C + c] = sum_m dY[m][o] A[m][tap][c] with A re-gathered as in the forward; a lane accumulates its
//             4 channels x 9 taps x O outputs over its pixels, groups / waves are summed through shuffles and LDS, workgroups
//             through a fixed-order second kernel.
// All three are bound by the gather's L1/L2 traffic (36 16-byte loads per 64 channels and pixel), not by HBM or MFMA.
#include <algorithm>

#include "eml_common.h"

namespace {

constexpr int kMaxC = 512;
constexpr int kNarrowBlocks = 1024;   // wgrad partials

__host__ __device__ inline bool narrow_supported(int C, int O) { return C >= 64 && C % 64 == 0 && C <= kMaxC && O >= 1 && O <= 4; }

// W2 [O][9C] -> LDS Wl[(tap * C/4 + c/4) * O + o] (float4 of 4 consecutive channels)
template <int O>
__device__ __forceinline__ void stage_weights(const float* __restrict__ W2, float4* Wl, int C) {
  const int C4 = C >> 2;
  for (int i = threadIdx.x; i < 9 * C4 * O; i += 256) {
    const int o = i % O, rest = i / O;
    const int c4 = rest % C4, t = rest / C4;
    Wl[i] = *reinterpret_cast<const float4*>(W2 + (size_t)o * 9 * C + (size_t)t * C + 4 * c4);
  }
  __syncthreads();
}

// the bilinear tap value of 4 consecutive channels: grid_sampler's corner order; corners off the map (index -1) add nothing.
// Loads are unconditional (clamped address, value selected afterwards): see sphere_conv_small.hip.
__device__ __forceinline__ float4 tap4(const float* __restrict__ xc, int C, const int4& id, const float4& w) {
  const float4 a = *reinterpret_cast<const float4*>(xc + (size_t)max(id.x, 0) * C);
  const float4 b = *reinterpret_cast<const float4*>(xc + (size_t)max(id.y, 0) * C);
  const float4 c = *reinterpret_cast<const float4*>(xc + (size_t)max(id.z, 0) * C);
  const float4 d = *reinterpret_cast<const float4*>(xc + (size_t)max(id.w, 0) * C);
  const float wa = id.x >= 0 ? w.x : 0.f, wb = id.y >= 0 ? w.y : 0.f, wc = id.z >= 0 ? w.z : 0.f, wd = id.w >= 0 ? w.w : 0.f;
  float4 v;
  v.x = a.x * wa; v.y = a.y * wa; v.z = a.z * wa; v.w = a.w * wa;
  v.x = fmaf(b.x, wb, v.x); v.y = fmaf(b.y, wb, v.y); v.z = fmaf(b.z, wb, v.z); v.w = fmaf(b.w, wb, v.w);
  v.x = fmaf(c.x, wc, v.x); v.y = fmaf(c.y, wc, v.y); v.z = fmaf(c.z, wc, v.z); v.w = fmaf(c.w, wc, v.w);
  v.x = fmaf(d.x, wd, v.x); v.y = fmaf(d.y, wd, v.y); v.z = fmaf(d.z, wd, v.z); v.w = fmaf(d.w, wd, v.w);
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}

template <int O>
__global__ __launch_bounds__(256) void sphere_conv_narrow_fwd_kernel(const float* __restrict__ X, const int* __restrict__ idx,
                                                                     const float* __restrict__ wgt,
                                                                     const float* __restrict__ W2, const float* __restrict__ bias,
                                                                     float* __restrict__ Y, int M, int HW, int Po, int C) {
  extern __shared__ __attribute__((aligned(16))) float4 Wl[];
  stage_weights<O>(W2, Wl, C);
  const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int C4 = C >> 2;
  for (int m = blockIdx.x * 16 + grp; m < M; m += gridDim.x * 16) {
    const int b = m / Po, p = m - b * Po;
    const float* xb = X + (size_t)b * HW * C;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    for (int c = 4 * l; c < C; c += 64) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int4 id = *reinterpret_cast<const int4*>(idx + ((size_t)p * 9 + t) * 4);
        const float4 w = *reinterpret_cast<const float4*>(wgt + ((size_t)p * 9 + t) * 4);
        const float4 a = tap4(xb + c, C, id, w);
        const float4* wl = Wl + (size_t)(t * C4 + (c >> 2)) * O;
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = dot4(a, wl[o], acc[o]);
      }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
      float v = acc[o];
      v += __shfl_xor(v, 8, 16);
      v += __shfl_xor(v, 4, 16);
      v += __shfl_xor(v, 2, 16);
      v += __shfl_xor(v, 1, 16);
      acc[o] = v;
    }
    if (l == 0) {
#pragma unroll
      for (int o = 0; o < O; ++o) Y[(size_t)m * O + o] = acc[o] + (bias ? bias[o] : 0.f);
    }
  }
}

// dX (B*HW, C): tidx / twgt (HW*9*ke) = the transposed tap table (see eml_sphere_conv_dgrad_fused_f32)
template <int O>
__global__ __launch_bounds__(256) void sphere_conv_narrow_dgrad_kernel(const float* __restrict__ dY, const int* __restrict__ tidx,
                                                                       const float* __restrict__ twgt, int ke,
                                                                       const float* __restrict__ W2, float* __restrict__ dX,
                                                                       int Min, int HW, int Po, int C) {
  extern __shared__ __attribute__((aligned(16))) float4 Wl[];
  stage_weights<O>(W2, Wl, C);
  const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int C4 = C >> 2;
  for (int qg0 = blockIdx.x * 16; qg0 < Min; qg0 += gridDim.x * 16) {   // uniform trip count: the shuffles need whole groups
    const int qg = qg0 + grp;
    const bool live = qg < Min;
    const int qc = live ? qg : Min - 1;
    const int b = qc / HW, q = qc - b * HW;
    float v[O];
#pragma unroll
    for (int o = 0; o < O; ++o) v[o] = 0.f;
    if (l < 9) {
      const int* ti = tidx + ((size_t)q * 9 + l) * ke;
      const float* tw = twgt + ((size_t)q * 9 + l) * ke;
      for (int s = 0; s < ke; ++s) {
        const int pi = ti[s];
        const float w = pi >= 0 ? tw[s] : 0.f;
        const float* g = dY + ((size_t)b * Po + max(pi, 0)) * O;
#pragma unroll
        for (int o = 0; o < O; ++o) v[o] = fmaf(w, g[o], v[o]);
      }
    }
    float vt[9][O];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int o = 0; o < O; ++o) vt[t][o] = __shfl(v[o], t, 16);
    if (live) {
      for (int c = 4 * l; c < C; c += 64) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float4* wl = Wl + (size_t)(t * C4 + (c >> 2)) * O;
#pragma unroll
          for (int o = 0; o < O; ++o) {
            const float4 wv = wl[o];
            d.x = fmaf(vt[t][o], wv.x, d.x); d.y = fmaf(vt[t][o], wv.y, d.y);
            d.z = fmaf(vt[t][o], wv.z, d.z); d.w = fmaf(vt[t][o], wv.w, d.w);
          }
        }
        *reinterpret_cast<float4*>(dX + (size_t)qg * C + c) = d;
      }
    }
  }
}

// partial[block][o][9C]; grid = (blocks, C / 64): block (x, y) sums pixels x*16 + grp, + gridDim.x*16, ... for channels 64y .. 64y+63
template <int O>
__global__ __launch_bounds__(256, 2) void sphere_conv_narrow_wgrad_kernel(const float* __restrict__ X, const int* __restrict__ idx,
                                                                       const float* __restrict__ wgt,
                                                                       const float* __restrict__ dY, float* __restrict__ partial,
                                                                       int M, int HW, int Po, int C) {
  __shared__ float red[4][16][9 * O * 4 + 1];
  const int l = threadIdx.x & 15, grp = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const int c = 64 * blockIdx.y + 4 * l;
  float4 aw[9][O];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < O; ++o) aw[t][o] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = blockIdx.x * 16 + grp; m < M; m += gridDim.x * 16) {
    const int b = m / Po, p = m - b * Po;
    const float* xc = X + (size_t)b * HW * C + c;
    float g[O];
#pragma unroll
    for (int o = 0; o < O; ++o) g[o] = dY[(size_t)m * O + o];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int4 id = *reinterpret_cast<const int4*>(idx + ((size_t)p * 9 + t) * 4);
      const float4 w = *reinterpret_cast<const float4*>(wgt + ((size_t)p * 9 + t) * 4);
      const float4 a = tap4(xc, C, id, w);
#pragma unroll
      for (int o = 0; o < O; ++o) {
        aw[t][o].x = fmaf(g[o], a.x, aw[t][o].x); aw[t][o].y = fmaf(g[o], a.y, aw[t][o].y);
        aw[t][o].z = fmaf(g[o], a.z, aw[t][o].z); aw[t][o].w = fmaf(g[o], a.w, aw[t][o].w);
      }
    }
  }
  // the 4 pixel groups of a wave (lanes l, l+16, l+32, l+48), then the 4 waves through LDS, in a fixed order
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < O; ++o) {
      float e[4] = {aw[t][o].x, aw[t][o].y, aw[t][o].z, aw[t][o].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = e[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if ((threadIdx.x & 63) < 16) red[wave][l][(t * O + o) * 4 + j] = s;
      }
    }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * O * 9 * C;
  for (int i = threadIdx.x; i < 16 * 9 * O * 4; i += 256) {
    const int ll = i / (9 * O * 4), k = i - ll * (9 * O * 4);
    const int t = k / (O * 4), o = (k / 4) % O, j = k & 3;
    const float s = (red[0][ll][k] + red[1][ll][k]) + (red[2][ll][k] + red[3][ll][k]);
    out[(size_t)o * 9 * C + (size_t)t * C + 64 * blockIdx.y + 4 * ll + j] = s;
  }
}

__global__ __launch_bounds__(256) void narrow_sum_partials_kernel(const float* __restrict__ partial, int S, size_t n,
                                                                  float* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += partial[(size_t)s * n + e];
  out[e] = t;
}

int narrow_wgrad_blocks(long M) { return (int)std::max<long>(1, std::min<long>(kNarrowBlocks, (M + 63) / 64)); }

#define EML_NARROW_DISPATCH(O_, STMT)      \
  switch (O_) {                            \
    case 1: { constexpr int OV = 1; STMT; } break; \
    case 2: { constexpr int OV = 2; STMT; } break; \
    case 3: { constexpr int OV = 3; STMT; } break; \
    default: { constexpr int OV = 4; STMT; } break; \
  }

}  // namespace

extern "C" int eml_sphere_conv_narrow_supported(int C, int O) { return narrow_supported(C, O) ? 1 : 0; }

extern "C" int eml_sphere_conv_narrow_fwd_f32(const float* X, const int* idx, const float* wgt, const float* W2,
                                              const float* bias, float* Y, int B, int HW, int Po, int C, int O,
                                              eml_stream_t stream) {
  if (!X || !idx || !wgt || !W2 || !Y || B < 0 || HW < 1 || Po < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_fwd_f32: null pointer or empty shape");
  if (!narrow_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_fwd_f32: need C %% 64 == 0, C <= %d, 1 <= O <= 4 (C=%d, O=%d)", kMaxC, C, O);
  const long M = (long)B * Po;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_fwd_f32: too many pixels");
  if (M == 0) return EML_OK;
  const size_t lds = (size_t)9 * C * O * sizeof(float);
  const int grid = (int)std::min<long>(8192, (M + 15) / 16);
  EML_NARROW_DISPATCH(O, {
    EML_ENSURE_LDS((&sphere_conv_narrow_fwd_kernel<OV>), lds);
    hipLaunchKernelGGL((sphere_conv_narrow_fwd_kernel<OV>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X, idx, wgt, W2, bias, Y,
                       (int)M, HW, Po, C);
  })
  return eml::check_launch("eml_sphere_conv_narrow_fwd_f32");
}

extern "C" int eml_sphere_conv_narrow_dgrad_f32(const float* dY, const int* tidx, const float* twgt, int ke, const float* W2,
                                                float* dX, int B, int HW, int Po, int C, int O, eml_stream_t stream) {
  if (!dY || !tidx || !twgt || !W2 || !dX || B < 0 || HW < 1 || Po < 1 || ke < 1 || ke > 8)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_dgrad_f32: null pointer, empty shape or ke outside 1..8");
  if (!narrow_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_dgrad_f32: need C %% 64 == 0, C <= %d, 1 <= O <= 4 (C=%d, O=%d)", kMaxC, C, O);
  const long Min = (long)B * HW;
  if (Min > 2147483647L || (long)B * Po > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_dgrad_f32: too many pixels");
  if (Min == 0) return EML_OK;
  const size_t lds = (size_t)9 * C * O * sizeof(float);
  const int grid = (int)std::min<long>(8192, (Min + 15) / 16);
  EML_NARROW_DISPATCH(O, {
    EML_ENSURE_LDS((&sphere_conv_narrow_dgrad_kernel<OV>), lds);
    hipLaunchKernelGGL((sphere_conv_narrow_dgrad_kernel<OV>), dim3(grid), dim3(256), lds, (hipStream_t)stream, dY, tidx, twgt, ke, W2,
                       dX, (int)Min, HW, Po, C);
  })
  return eml::check_launch("eml_sphere_conv_narrow_dgrad_f32");
}

extern "C" size_t eml_sphere_conv_narrow_wgrad_partial_floats(int B, int Po, int C, int O) {
  if (!narrow_supported(C, O) || B < 1 || Po < 1) return 0;
  return (size_t)narrow_wgrad_blocks((long)B * Po) * O * 9 * C;
}

extern "C" int eml_sphere_conv_narrow_wgrad_f32(const float* X, const int* idx, const float* wgt, const float* dY, float* partial,
                                                float* dW2, int B, int HW, int Po, int C, int O, eml_stream_t stream) {
  if (!X || !idx || !wgt || !dY || !partial || !dW2 || B < 1 || HW < 1 || Po < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_wgrad_f32: null pointer or empty shape");
  if (!narrow_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_wgrad_f32: need C %% 64 == 0, C <= %d, 1 <= O <= 4 (C=%d, O=%d)", kMaxC, C, O);
  const long M = (long)B * Po;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_wgrad_f32: too many pixels");
  const int blocks = narrow_wgrad_blocks(M);
  const dim3 grid(blocks, C / 64);
  EML_NARROW_DISPATCH(O, {
    hipLaunchKernelGGL((sphere_conv_narrow_wgrad_kernel<OV>), grid, dim3(256), 0, (hipStream_t)stream, X, idx, wgt, dY, partial, (int)M,
                       HW, Po, C);
  })
  int rc = eml::check_launch("eml_sphere_conv_narrow_wgrad_f32");
  if (rc) return rc;
  const size_t n = (size_t)O * 9 * C;
  hipLaunchKernelGGL(narrow_sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, blocks,
                     n, dW2);
  return eml::check_launch("eml_sphere_conv_narrow_wgrad_f32(reduce)");
}
