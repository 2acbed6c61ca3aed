// SphereConv2D support kernels for the GenProjector (reference models/networks/spherenet/sphere_cnn.py:111-124):
//     x -> grid_sample(x, fixed tangent-plane grid) -> (B, C, 3H', 3W') -> conv2d(stride 3)
// is restated as  A9 = im2col_sphere(x)  (B*H'*W' rows, 9*C columns, pixel-major / channels-last),
// Y = A9 * W2^T (library GEMM), and in the backward  dA9 = dY * W2,  dW2 = dY^T * A9,  dx = col2im_sphere(dA9).
//
// Why: on stock ops 75 % of the projector step is ATen's grid_sampler (65 % its backward, an atomicAdd scatter
// over the 9x blown-up tensor, profiles/r01_projector_stock_kernel_stats.csv).  The sampling pattern depends only
// on (H, W, stride), so its transpose is a fixed sparse matrix: the backward here is a deterministic CSR gather
// (no atomics), and the forward gather writes the GEMM operand directly in the layout the GEMM wants (no
// NCHW<->NHWC transposes).
//
//   tap table : for every (output pixel, tap) the 4 bilinear corners (input pixel index or -1) and weights,
//               computed ONCE per geometry with exactly grid_sample's arithmetic (align_corners = False, zero padding)
//   im2col    : A9[(b*Po + p)*9 + tap][c] = sum_k wgt[p,tap,k] * X[b][idx[p,tap,k]][c]
//   col2im    : dX[b][q][c] = sum_{e in row q of the CSR transpose} w_e * dA9[(b*Po*9 + src_e)][c]
// All three are HBM-streaming kernels (lanes along the contiguous channel axis, 16-byte accesses when C % 4 == 0).
#include <algorithm>

#include "eml_common.h"

namespace {

// ATen grid_sampler_unnormalize, align_corners = false (GridSampler.cuh): ((coord + 1) * size - 1) / 2
__device__ __forceinline__ float unnormalize(float coord, int size) { return ((coord + 1.f) * size - 1.f) / 2.f; }

__global__ __launch_bounds__(256) void sphere_tap_table_kernel(const float* __restrict__ grid, int H, int W, int Ho, int Wo,
                                                               int* __restrict__ idx, float* __restrict__ wgt) {
  const int total = Ho * Wo * 9;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int p = e / 9, tap = e - 9 * p;
    const int r = p / Wo, c = p - r * Wo;
    const int a = tap / 3, b = tap - 3 * a;
    const float* g = grid + ((size_t)(3 * r + a) * (3 * Wo) + (3 * c + b)) * 2;
    const float ix = unnormalize(g[0], W), iy = unnormalize(g[1], H);
    // corners and weights exactly as grid_sampler_2d (bilinear): nw, ne, sw, se
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float xe = fx + 1.f, ye = fy + 1.f;
    const float w4[4] = {(xe - ix) * (ye - iy), (ix - fx) * (ye - iy), (xe - ix) * (iy - fy), (ix - fx) * (iy - fy)};
    const int xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = xs[k] >= 0 && xs[k] < W && ys[k] >= 0 && ys[k] < H;
      idx[(size_t)e * 4 + k] = in ? ys[k] * W + xs[k] : -1;
      wgt[(size_t)e * 4 + k] = in ? w4[k] : 0.f;
    }
  }
}

// Thread layout of the two gather kernels: TPR (a power of two) consecutive threads share one row and stride over
// its channels, blockIdx.y is the batch sample -- all index arithmetic is 32-bit with divisions by constants only
// (a flat 64-bit index with three runtime divisions per element made the first version VALU-bound at 2.1 TB/s).
template <int VEC, int TPR>
__global__ __launch_bounds__(256) void sphere_im2col_kernel(const float* __restrict__ X, const int* __restrict__ idx,
                                                            const float* __restrict__ wgt, float* __restrict__ A9,
                                                            int HW, int Po, int C) {
  constexpr int RPB = 256 / TPR;  // rows per block per pass
  const int cv = C / VEC, nrows = Po * 9;
  const int rl = threadIdx.x / TPR, cl = threadIdx.x % TPR;
  const float* xb = X + (size_t)blockIdx.y * HW * C;
  float* ab = A9 + (size_t)blockIdx.y * nrows * C;
  for (int row = blockIdx.x * RPB + rl; row < nrows; row += gridDim.x * RPB) {
    const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)row * 4);   // row = p*9 + tap: the table's own order
    const float4 w = *reinterpret_cast<const float4*>(wgt + (size_t)row * 4);
    const int ids[4] = {id.x, id.y, id.z, id.w};
    const float ws[4] = {w.x, w.y, w.z, w.w};
    for (int c = cl; c < cv; c += TPR) {
      if constexpr (VEC == 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // same accumulation order as grid_sampler_2d: nw, ne, sw, se; a corner off the map adds +0 (its load is issued anyway,
        // at a clamped address: under `if (id >= 0)` each corner was its own basic block -- load, wait, add -- four serial
        // round trips per element)
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(xb + (size_t)max(ids[k], 0) * C + 4 * c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // a corner off the map contributes exactly +0 whatever the clamped address holds (0 * Inf would be NaN where
          // grid_sample's zero padding is finite: ADVICE round 4) -- the kernel is bound by its stores, the selects are free
          const bool in = ids[k] >= 0;
          const float wk = in ? ws[k] : 0.f;
          acc.x += (in ? v[k].x : 0.f) * wk;
          acc.y += (in ? v[k].y : 0.f) * wk;
          acc.z += (in ? v[k].z : 0.f) * wk;
          acc.w += (in ? v[k].w : 0.f) * wk;
        }
        // streaming store: A9 is 9x the input and is next read by the GEMM long after it has left L2
        float* dst = ab + (size_t)row * C + 4 * c;
        __builtin_nontemporal_store(acc.x, dst);
        __builtin_nontemporal_store(acc.y, dst + 1);
        __builtin_nontemporal_store(acc.z, dst + 2);
        __builtin_nontemporal_store(acc.w, dst + 3);
      } else {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (ids[k] >= 0) acc += xb[(size_t)ids[k] * C + c] * ws[k];
        ab[(size_t)row * C + c] = acc;
      }
    }
  }
}

template <int VEC, int TPR>
__global__ __launch_bounds__(256) void sphere_col2im_kernel(const float* __restrict__ dA9, const int* __restrict__ ptr,
                                                            const int* __restrict__ src, const float* __restrict__ w,
                                                            float* __restrict__ dX, int HW, int Po, int C) {
  constexpr int RPB = 256 / TPR;
  const int cv = C / VEC;
  const int rl = threadIdx.x / TPR, cl = threadIdx.x % TPR;
  const float* ab = dA9 + (size_t)blockIdx.y * Po * 9 * C;
  float* xb = dX + (size_t)blockIdx.y * HW * C;
  for (int q = blockIdx.x * RPB + rl; q < HW; q += gridDim.x * RPB) {
    const int k0 = ptr[q], k1 = ptr[q + 1];
    for (int c = cl; c < cv; c += TPR) {
      if constexpr (VEC == 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // four entries per step: their (source row, weight) loads first, then the four row reads, then the FMAs in CSR
        // order (the sum's order is unchanged).  One entry per iteration was two dependent memory latencies x ~36 entries
        // per input pixel: the kernel ran at 1.1 TB/s on a tensor it reads exactly once.
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
          const int s0 = src[k], s1 = src[k + 1], s2 = src[k + 2], s3 = src[k + 3];
          const float w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
          const float4 v0 = *reinterpret_cast<const float4*>(ab + (size_t)s0 * C + 4 * c);
          const float4 v1 = *reinterpret_cast<const float4*>(ab + (size_t)s1 * C + 4 * c);
          const float4 v2 = *reinterpret_cast<const float4*>(ab + (size_t)s2 * C + 4 * c);
          const float4 v3 = *reinterpret_cast<const float4*>(ab + (size_t)s3 * C + 4 * c);
          acc.x = fmaf(w0, v0.x, acc.x); acc.y = fmaf(w0, v0.y, acc.y); acc.z = fmaf(w0, v0.z, acc.z); acc.w = fmaf(w0, v0.w, acc.w);
          acc.x = fmaf(w1, v1.x, acc.x); acc.y = fmaf(w1, v1.y, acc.y); acc.z = fmaf(w1, v1.z, acc.z); acc.w = fmaf(w1, v1.w, acc.w);
          acc.x = fmaf(w2, v2.x, acc.x); acc.y = fmaf(w2, v2.y, acc.y); acc.z = fmaf(w2, v2.z, acc.z); acc.w = fmaf(w2, v2.w, acc.w);
          acc.x = fmaf(w3, v3.x, acc.x); acc.y = fmaf(w3, v3.y, acc.y); acc.z = fmaf(w3, v3.z, acc.z); acc.w = fmaf(w3, v3.w, acc.w);
        }
        for (; k < k1; ++k) {
          const float wk = w[k];
          const float4 v = *reinterpret_cast<const float4*>(ab + (size_t)src[k] * C + 4 * c);
          acc.x = fmaf(wk, v.x, acc.x);
          acc.y = fmaf(wk, v.y, acc.y);
          acc.z = fmaf(wk, v.z, acc.z);
          acc.w = fmaf(wk, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(xb + (size_t)q * C + 4 * c) = acc;
      } else {
        float acc = 0.f;
        int k = k0;
        for (; k + 4 <= k1; k += 4) {   // as above: the four entries' loads first, the FMAs in CSR order
          const int s0 = src[k], s1 = src[k + 1], s2 = src[k + 2], s3 = src[k + 3];
          const float w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
          const float v0 = ab[(size_t)s0 * C + c], v1 = ab[(size_t)s1 * C + c], v2 = ab[(size_t)s2 * C + c], v3 = ab[(size_t)s3 * C + c];
          acc = fmaf(w3, v3, fmaf(w2, v2, fmaf(w1, v1, fmaf(w0, v0, acc))));
        }
        for (; k < k1; ++k) acc = fmaf(w[k], ab[(size_t)src[k] * C + c], acc);
        xb[(size_t)q * C + c] = acc;
      }
    }
  }
}

inline int stream_grid(size_t total) {
  const size_t g = (total + 255) / 256;
  return (int)(g < 16384 ? (g ? g : 1) : 16384);
}

}  // namespace

extern "C" int eml_sphere_tap_table_f32(const float* grid, int H, int W, int Ho, int Wo, int* idx, float* wgt,
                                        eml_stream_t stream) {
  if (!grid || !idx || !wgt || H < 1 || W < 1 || Ho < 1 || Wo < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_tap_table_f32: bad arguments");
  hipLaunchKernelGGL(sphere_tap_table_kernel, dim3(stream_grid((size_t)Ho * Wo * 9)), dim3(256), 0, (hipStream_t)stream,
                     grid, H, W, Ho, Wo, idx, wgt);
  return eml::check_launch("eml_sphere_tap_table_f32");
}

// threads per row = the smallest power of two >= C / VEC (capped at 256): no idle lanes in the channel loop
#define EML_SPHERE_CASE(KERNEL, V, T, ...) \
  case T: hipLaunchKernelGGL((KERNEL<V, T>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;
#define EML_SPHERE_LAUNCH(KERNEL, rows, ...)                                                             \
  do {                                                                                                  \
    const int cv = (C % 4 == 0) ? C / 4 : C;                                                            \
    int tpr = 4;                                                                                        \
    while (tpr < cv && tpr < 256) tpr <<= 1;                                                            \
    const int gx = (int)std::min<size_t>(((size_t)(rows) + (256 / tpr) - 1) / (256 / tpr), (size_t)8192); \
    const dim3 grid(gx, B);                                                                             \
    if (C % 4 == 0) {                                                                                   \
      switch (tpr) {                                                                                    \
        EML_SPHERE_CASE(KERNEL, 4, 4, __VA_ARGS__)                                                      \
        EML_SPHERE_CASE(KERNEL, 4, 8, __VA_ARGS__)                                                      \
        EML_SPHERE_CASE(KERNEL, 4, 16, __VA_ARGS__)                                                     \
        EML_SPHERE_CASE(KERNEL, 4, 32, __VA_ARGS__)                                                     \
        EML_SPHERE_CASE(KERNEL, 4, 64, __VA_ARGS__)                                                     \
        EML_SPHERE_CASE(KERNEL, 4, 128, __VA_ARGS__)                                                    \
        EML_SPHERE_CASE(KERNEL, 4, 256, __VA_ARGS__)                                                    \
      }                                                                                                 \
    } else {                                                                                            \
      switch (tpr) {                                                                                    \
        EML_SPHERE_CASE(KERNEL, 1, 4, __VA_ARGS__)                                                      \
        EML_SPHERE_CASE(KERNEL, 1, 8, __VA_ARGS__)                                                      \
        EML_SPHERE_CASE(KERNEL, 1, 16, __VA_ARGS__)                                                     \
        EML_SPHERE_CASE(KERNEL, 1, 32, __VA_ARGS__)                                                     \
        EML_SPHERE_CASE(KERNEL, 1, 64, __VA_ARGS__)                                                     \
        EML_SPHERE_CASE(KERNEL, 1, 128, __VA_ARGS__)                                                    \
        EML_SPHERE_CASE(KERNEL, 1, 256, __VA_ARGS__)                                                    \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)

extern "C" int eml_sphere_im2col_f32(const float* X, const int* idx, const float* wgt, float* A9, int B, int HW, int Po,
                                     int C, eml_stream_t stream) {
  if (!X || !idx || !wgt || !A9 || B < 0 || HW < 1 || Po < 1 || C < 1 || B > 65535)
    return eml::fail(EML_EINVAL, "eml_sphere_im2col_f32: bad arguments");
  if (B == 0) return EML_OK;
  EML_SPHERE_LAUNCH(sphere_im2col_kernel, (size_t)Po * 9, X, idx, wgt, A9, HW, Po, C);
  return eml::check_launch("eml_sphere_im2col_f32");
}

extern "C" int eml_sphere_col2im_f32(const float* dA9, const int* ptr, const int* src, const float* w, float* dX, int B,
                                     int HW, int Po, int C, eml_stream_t stream) {
  if (!dA9 || !ptr || !src || !w || !dX || B < 0 || HW < 1 || Po < 1 || C < 1 || B > 65535)
    return eml::fail(EML_EINVAL, "eml_sphere_col2im_f32: bad arguments");
  if (B == 0) return EML_OK;
  EML_SPHERE_LAUNCH(sphere_col2im_kernel, (size_t)HW, dA9, ptr, src, w, dX, HW, Po, C);
  return eml::check_launch("eml_sphere_col2im_f32");
}
