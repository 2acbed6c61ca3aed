// InstanceNorm2d(affine=False) + the LeakyReLU that follows it, forward and backward, for the PatchGAN discriminator and the
// crop encoder of the GenProjector (reference models/networks/normalization.py:20-62 'spectralinstance',
// discriminator.py:84-98, generator.py:96-123: norm_layer(conv) then LeakyReLU(0.2)).
//
// On stock ops an instance norm of a channels-last tensor is: a copy to NCHW, batch_norm over B*C "channels" of a few hundred
// to a few thousand elements each (ATen's collect_statistics / transform / backward kernels launch one block per channel:
// ~100 us for 17-67 MB tensors), a copy back, and the activation as one more pass -- five to six launches each way.
// Here: one launch each way.  A workgroup owns (sample, 32 channels) of a pixel-major tensor (or one (sample, channel) plane
// of an NCHW one): pass 1 accumulates the statistics in f64, pass 2 re-reads its slab (<= 256 KB: L2 / MALL resident),
// normalises, applies the activation and writes.  (mean, istd) are kept for the backward:
//     g' = gy * act'(xhat),   dx = istd * (g' - mean(g') - xhat * mean(g' xhat))          per (sample, channel).
#include "eml_common.h"

namespace {

__device__ __forceinline__ float lrelu(float t, float slope) { return t > 0.f ? t : slope * t; }

// ---------------------------------------------------------------------------------------------- pixel-major (NHWC)
// block (8 x 32): tx = 4-channel group of the 32-channel chunk, ty = pixel lane.  red: [32][8][8] doubles.
template <bool BWD>
__global__ __launch_bounds__(256) void instance_norm_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 float* __restrict__ out, float* __restrict__ stats,
                                                                 int HW, int C, float eps, float slope) {
  __shared__ double red[32][8][8];
  __shared__ float par[2][32];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int b = blockIdx.y, c = blockIdx.x * 32 + 4 * tx;
  const bool live = c < C;
  const size_t base = (size_t)b * HW * C + c;
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
  if constexpr (BWD) {
    if (live) {
      const float* s = stats + ((size_t)b * C + c) * 2;
      const float4 s0 = *reinterpret_cast<const float4*>(s), s1 = *reinterpret_cast<const float4*>(s + 4);
      mu = make_float4(s0.x, s0.z, s1.x, s1.z);
      is = make_float4(s0.y, s0.w, s1.y, s1.w);
    }
  }
  double a[4] = {0., 0., 0., 0.}, q[4] = {0., 0., 0., 0.};
  if (live) {
#pragma unroll 4
    for (int p = ty; p < HW; p += 32) {
      const float4 v = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
      if constexpr (!BWD) {
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
        q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
      } else {
        const float4 g = *reinterpret_cast<const float4*>(gy + base + (size_t)p * C);
        const float h0 = (v.x - mu.x) * is.x, h1 = (v.y - mu.y) * is.y, h2 = (v.z - mu.z) * is.z, h3 = (v.w - mu.w) * is.w;
        const float g0 = h0 > 0.f ? g.x : slope * g.x, g1 = h1 > 0.f ? g.y : slope * g.y;
        const float g2 = h2 > 0.f ? g.z : slope * g.z, g3 = h3 > 0.f ? g.w : slope * g.w;
        a[0] += g0; a[1] += g1; a[2] += g2; a[3] += g3;
        q[0] += (double)g0 * h0; q[1] += (double)g1 * h1; q[2] += (double)g2 * h2; q[3] += (double)g3 * h3;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[ty][tx][k] = a[k]; red[ty][tx][4 + k] = q[k]; }
  __syncthreads();
  if (threadIdx.x < 64) {   // thread -> (tx' = id / 8, component id % 8): sum over the 32 pixel lanes
    const int txx = threadIdx.x >> 3, k = threadIdx.x & 7;
    double s = 0.;
    for (int r = 0; r < 32; ++r) s += red[r][txx][k];
    red[0][txx][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 32) {   // one thread per channel of the chunk
    const int txx = threadIdx.x >> 2, k = threadIdx.x & 3;
    const double s = red[0][txx][k], t = red[0][txx][4 + k], n = (double)HW;
    if constexpr (!BWD) {
      const double m = s / n;
      double var = t / n - m * m;   // biased, as F.instance_norm / batch_norm(training) normalise
      var = var > 0. ? var : 0.;
      const float istd = (float)(1.0 / sqrt(var + (double)eps));
      par[0][threadIdx.x] = (float)m;
      par[1][threadIdx.x] = istd;
      const int cc = blockIdx.x * 32 + threadIdx.x;
      if (cc < C) {
        stats[((size_t)b * C + cc) * 2] = (float)m;
        stats[((size_t)b * C + cc) * 2 + 1] = istd;
      }
    } else {
      par[0][threadIdx.x] = (float)(s / n);
      par[1][threadIdx.x] = (float)(t / n);
    }
  }
  __syncthreads();
  if (!live) return;
  const float4 p0 = make_float4(par[0][4 * tx], par[0][4 * tx + 1], par[0][4 * tx + 2], par[0][4 * tx + 3]);
  const float4 p1 = make_float4(par[1][4 * tx], par[1][4 * tx + 1], par[1][4 * tx + 2], par[1][4 * tx + 3]);
#pragma unroll 4
  for (int p = ty; p < HW; p += 32) {
    const float4 v = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
    float4 o;
    if constexpr (!BWD) {
      o.x = lrelu((v.x - p0.x) * p1.x, slope);
      o.y = lrelu((v.y - p0.y) * p1.y, slope);
      o.z = lrelu((v.z - p0.z) * p1.z, slope);
      o.w = lrelu((v.w - p0.w) * p1.w, slope);
    } else {
      const float4 g = *reinterpret_cast<const float4*>(gy + base + (size_t)p * C);
      const float h0 = (v.x - mu.x) * is.x, h1 = (v.y - mu.y) * is.y, h2 = (v.z - mu.z) * is.z, h3 = (v.w - mu.w) * is.w;
      o.x = is.x * ((h0 > 0.f ? g.x : slope * g.x) - p0.x - h0 * p1.x);
      o.y = is.y * ((h1 > 0.f ? g.y : slope * g.y) - p0.y - h1 * p1.y);
      o.z = is.z * ((h2 > 0.f ? g.z : slope * g.z) - p0.z - h2 * p1.z);
      o.w = is.w * ((h3 > 0.f ? g.w : slope * g.w) - p0.w - h3 * p1.w);
    }
    *reinterpret_cast<float4*>(out + base + (size_t)p * C) = o;
  }
}

// ---------------------------------------------------------------------------------------------- planar (NCHW)
// one workgroup per (sample, channel) plane of HW contiguous floats
template <bool BWD>
__global__ __launch_bounds__(256) void instance_norm_nchw_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 float* __restrict__ out, float* __restrict__ stats,
                                                                 int HW, float eps, float slope) {
  __shared__ double red[2][4];
  __shared__ float par[2];
  const size_t plane = blockIdx.x;
  const float* xp = x + plane * HW;
  const float* gp = BWD ? gy + plane * HW : nullptr;
  float mu = 0.f, is = 0.f;
  if constexpr (BWD) { mu = stats[plane * 2]; is = stats[plane * 2 + 1]; }
  double a = 0., q = 0.;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float v = xp[p];
    if constexpr (!BWD) {
      a += v;
      q += (double)v * v;
    } else {
      const float h = (v - mu) * is, g = h > 0.f ? gp[p] : slope * gp[p];
      a += g;
      q += (double)g * h;
    }
  }
  for (int off = 32; off; off >>= 1) {
    a += __shfl_xor(a, off);
    q += __shfl_xor(q, off);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = red[0][0] + red[0][1] + red[0][2] + red[0][3], t = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double n = (double)HW;
    if constexpr (!BWD) {
      const double m = s / n;
      double var = t / n - m * m;
      var = var > 0. ? var : 0.;
      par[0] = (float)m;
      par[1] = (float)(1.0 / sqrt(var + (double)eps));
      stats[plane * 2] = par[0];
      stats[plane * 2 + 1] = par[1];
    } else {
      par[0] = (float)(s / n);
      par[1] = (float)(t / n);
    }
  }
  __syncthreads();
  const float p0 = par[0], p1 = par[1];
  float* op = out + plane * HW;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float v = xp[p];
    if constexpr (!BWD) {
      op[p] = lrelu((v - p0) * p1, slope);
    } else {
      const float h = (v - mu) * is, g = h > 0.f ? gp[p] : slope * gp[p];
      op[p] = is * (g - p0 - h * p1);
    }
  }
}

int check_args(const char* what, const void* a, const void* b, const void* c, int B, int HW, int C, int channels_last,
               float slope) {
  if (!a || !b || !c || B < 0 || HW < 1 || C < 1) return eml::fail(EML_EINVAL, "%s: null pointer or empty shape", what);
  if (channels_last && (C % 4)) return eml::fail(EML_EINVAL, "%s: pixel-major layout needs C %% 4 == 0 (C=%d)", what, C);
  if (!(slope >= 0.f && slope <= 1.f)) return eml::fail(EML_EINVAL, "%s: slope %g outside [0, 1]", what, (double)slope);
  if (B > 65535 || (long)B * C > 2147483647L) return eml::fail(EML_EINVAL, "%s: batch too large", what);
  return EML_OK;
}

}  // namespace

extern "C" int eml_instance_norm_act_fwd_f32(const float* x, float* y, float* stats, int B, int HW, int C,
                                             int channels_last, float eps, float slope, eml_stream_t stream) {
  if (int rc = check_args("eml_instance_norm_act_fwd_f32", x, y, stats, B, HW, C, channels_last, slope)) return rc;
  if (B == 0) return EML_OK;
  if (channels_last)
    hipLaunchKernelGGL(instance_norm_nhwc_kernel<false>, dim3((C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, x, nullptr,
                       y, stats, HW, C, eps, slope);
  else
    hipLaunchKernelGGL(instance_norm_nchw_kernel<false>, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, x,
                       nullptr, y, stats, HW, eps, slope);
  return eml::check_launch("eml_instance_norm_act_fwd_f32");
}

extern "C" int eml_instance_norm_act_bwd_f32(const float* gy, const float* x, const float* stats, float* dx, int B, int HW,
                                             int C, int channels_last, float slope, eml_stream_t stream) {
  if (int rc = check_args("eml_instance_norm_act_bwd_f32", gy, x, stats, B, HW, C, channels_last, slope)) return rc;
  if (!dx) return eml::fail(EML_EINVAL, "eml_instance_norm_act_bwd_f32: null dx");
  if (B == 0) return EML_OK;
  if (channels_last)
    hipLaunchKernelGGL(instance_norm_nhwc_kernel<true>, dim3((C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, x, gy, dx,
                       const_cast<float*>(stats), HW, C, 0.f, slope);
  else
    hipLaunchKernelGGL(instance_norm_nchw_kernel<true>, dim3((unsigned)(B * C)), dim3(256), 0, (hipStream_t)stream, x, gy, dx,
                       const_cast<float*>(stats), HW, 0.f, slope);
  return eml::check_launch("eml_instance_norm_act_bwd_f32");
}
