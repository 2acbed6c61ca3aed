// SPADE modulation of the GenProjector (reference models/networks/normalization.py:101-115, architecture.py:49-62):
//     out = leaky_relu( normalized * (1 + gamma) + beta, slope )       (slope = 1: no activation, the shortcut branch)
// On stock ops this is add-scalar, mul, add, leaky_relu -- four elementwise launches and ten tensor traversals --
// and as many again in the backward, plus a split / cat of the (gamma | beta) tensor the fused SphereConv produces.
// Here: one streaming pass each way over pixel-major (channels-last) rows, 16-byte accesses; gamma and beta are
// read from, and their gradients written to, the two channel halves of ONE (rows, 2C) tensor.
#include "eml_common.h"

namespace {

__device__ __forceinline__ float lrelu(float t, float slope) { return t > 0.f ? t : slope * t; }

__global__ __launch_bounds__(256) void spade_modulate_fwd_kernel(const float* __restrict__ xn, int ld_x,
                                                                 const float* __restrict__ gb, int ld_gb,
                                                                 float* __restrict__ y, int ld_y, int rows, int C,
                                                                 float slope) {
  const int cv = C >> 2;
  const size_t total = (size_t)rows * cv;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / cv;
    const int c = (int)(e - row * cv) * 4;
    const float4 a = *reinterpret_cast<const float4*>(xn + row * ld_x + c);
    const float4 g = *reinterpret_cast<const float4*>(gb + row * ld_gb + c);
    const float4 b = *reinterpret_cast<const float4*>(gb + row * ld_gb + C + c);
    float4 o;
    o.x = lrelu(fmaf(a.x, 1.f + g.x, b.x), slope);
    o.y = lrelu(fmaf(a.y, 1.f + g.y, b.y), slope);
    o.z = lrelu(fmaf(a.z, 1.f + g.z, b.z), slope);
    o.w = lrelu(fmaf(a.w, 1.f + g.w, b.w), slope);
    *reinterpret_cast<float4*>(y + row * ld_y + c) = o;
  }
}

__global__ __launch_bounds__(256) void spade_modulate_bwd_kernel(const float* __restrict__ gy, int ld_gy,
                                                                 const float* __restrict__ xn, int ld_x,
                                                                 const float* __restrict__ gb, int ld_gb,
                                                                 float* __restrict__ dxn, int ld_dx,
                                                                 float* __restrict__ dgb, int ld_dgb, int rows, int C,
                                                                 float slope) {
  const int cv = C >> 2;
  const size_t total = (size_t)rows * cv;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / cv;
    const int c = (int)(e - row * cv) * 4;
    const float4 u = *reinterpret_cast<const float4*>(gy + row * ld_gy + c);
    const float4 a = *reinterpret_cast<const float4*>(xn + row * ld_x + c);
    const float4 g = *reinterpret_cast<const float4*>(gb + row * ld_gb + c);
    const float4 b = *reinterpret_cast<const float4*>(gb + row * ld_gb + C + c);
    float4 d, dx, dg;
    // leaky_relu'(t) = 1 for t > 0, slope otherwise (ATen's leaky_relu_backward)
    d.x = fmaf(a.x, 1.f + g.x, b.x) > 0.f ? u.x : slope * u.x;
    d.y = fmaf(a.y, 1.f + g.y, b.y) > 0.f ? u.y : slope * u.y;
    d.z = fmaf(a.z, 1.f + g.z, b.z) > 0.f ? u.z : slope * u.z;
    d.w = fmaf(a.w, 1.f + g.w, b.w) > 0.f ? u.w : slope * u.w;
    dx.x = d.x * (1.f + g.x);
    dx.y = d.y * (1.f + g.y);
    dx.z = d.z * (1.f + g.z);
    dx.w = d.w * (1.f + g.w);
    dg.x = d.x * a.x;
    dg.y = d.y * a.y;
    dg.z = d.z * a.z;
    dg.w = d.w * a.w;
    *reinterpret_cast<float4*>(dxn + row * ld_dx + c) = dx;
    *reinterpret_cast<float4*>(dgb + row * ld_dgb + c) = dg;
    *reinterpret_cast<float4*>(dgb + row * ld_dgb + C + c) = d;
  }
}

inline int grid_for(size_t n4) {
  const size_t g = (n4 + 255) / 256;
  return (int)(g < 8192 ? (g ? g : 1) : 8192);
}

inline bool bad_ld(int ld, int need) { return ld < need || (ld & 3); }

}  // namespace

extern "C" int eml_spade_modulate_fwd_f32(const float* xn, int ld_x, const float* gb, int ld_gb, float* y, int ld_y,
                                          long rows, int C, float slope, eml_stream_t stream) {
  if (!xn || !gb || !y || rows < 0 || rows > 2147483647L || C < 4 || (C & 3) || bad_ld(ld_x, C) || bad_ld(ld_gb, 2 * C) ||
      bad_ld(ld_y, C))
    return eml::fail(EML_EINVAL, "eml_spade_modulate_fwd_f32: bad arguments");
  if (rows == 0) return EML_OK;
  hipLaunchKernelGGL(spade_modulate_fwd_kernel, dim3(grid_for((size_t)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     xn, ld_x, gb, ld_gb, y, ld_y, (int)rows, C, slope);
  return eml::check_launch("eml_spade_modulate_fwd_f32");
}

extern "C" int eml_spade_modulate_bwd_f32(const float* gy, int ld_gy, const float* xn, int ld_x, const float* gb,
                                          int ld_gb, float* dxn, int ld_dx, float* dgb, int ld_dgb, long rows, int C,
                                          float slope, eml_stream_t stream) {
  if (!gy || !xn || !gb || !dxn || !dgb || rows < 0 || rows > 2147483647L || C < 4 || (C & 3) || bad_ld(ld_gy, C) ||
      bad_ld(ld_x, C) || bad_ld(ld_gb, 2 * C) || bad_ld(ld_dx, C) || bad_ld(ld_dgb, 2 * C))
    return eml::fail(EML_EINVAL, "eml_spade_modulate_bwd_f32: bad arguments");
  if (rows == 0) return EML_OK;
  hipLaunchKernelGGL(spade_modulate_bwd_kernel, dim3(grid_for((size_t)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     gy, ld_gy, xn, ld_x, gb, ld_gb, dxn, ld_dx, dgb, ld_dgb, (int)rows, C, slope);
  return eml::check_launch("eml_spade_modulate_bwd_f32");
}
