// SPADE modulation of the GenProjector (reference models/networks/normalization.py:101-115, architecture.py:49-62):
//     out = leaky_relu( normalized * (1 + gamma) + beta, slope )       (slope = 1: no activation, the shortcut branch)
// On stock ops this is add-scalar, mul, add, leaky_relu -- four elementwise launches and ten tensor traversals --
// and as many again in the backward, plus a split / cat of the (gamma | beta) tensor the fused SphereConv produces.
// Here: one streaming pass each way over pixel-major (channels-last) rows, 16-byte accesses; gamma and beta are
// read from, and their gradients written to, the two channel halves of ONE (rows, 2C) tensor.
//
// SPADE's parameter-free BatchNorm (normalization.py:86-104; sync_batchnorm/batchnorm.py:105-126 across replicas) is
// folded into the same passes: `bn_stats` reads x once for the per-channel (sum, sum of squares) in f64 partials,
// the modulation normalises inline ((x - mean) * istd never exists in memory), the backward pass emits the partial
// sums (sum dxn, sum dxn*xhat) of BatchNorm's backward in its epilogue, and `bn_bwd_apply` finishes
// dx = istd * (dxn - S1/n - xhat * S2/n).  Between partials and finalisation the host can all-reduce the (2C+1)-float
// sums across ranks (one process per GPU): that is the whole of SynchronizedBatchNorm.
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

// ---- fixed-column layout of the reducing kernels: thread -> (row lane, 4-channel group); cvp = the power of two
// >= C/4 (<= 256); a block covers 256/cvp rows per pass.  Channel groups beyond 256*4 channels loop (cg += 256).
struct ColMap {
  int cv, cvp, rpp, rl, cl;
  __device__ ColMap(int C) {
    cv = C >> 2;
    cvp = 1;
    while (cvp < cv && cvp < 256) cvp <<= 1;
    rpp = 256 / cvp;
    rl = threadIdx.x / cvp;
    cl = threadIdx.x % cvp;
  }
};

// block-level reduction of per-thread f64 pairs over the row lanes, then one partial row per block
template <int NV>
__device__ __forceinline__ void reduce_rows_and_store(const ColMap& m, int cg, double (&v)[NV], double* red /*[256][NV]*/,
                                                      double* __restrict__ out /*[C][NV/4]... see callers*/, int C) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) red[threadIdx.x * NV + k] = v[k];
  __syncthreads();
  if (m.rl == 0 && cg < m.cv) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double t = red[m.cl * NV + k];
      for (int r = 1; r < m.rpp; ++r) t += red[(r * m.cvp + m.cl) * NV + k];
      // v = {s1[0..3], s2[0..3] (, s3[0..3], s4[0..3])}: channel 4*cg + (k & 3), statistic k >> 2; statistics 2, 3 form a
      // second (C, 2) table behind the first and its count slot
      const int st = k >> 2;
      out[(st >> 1) * (size_t)(2 * C + 1) + (size_t)(4 * cg + (k & 3)) * 2 + (st & 1)] = t;
    }
  }
}

// partials[block][c][2] = (sum x, sum x^2) over this block's rows, accumulated per element in f64
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int ld, int rows, int C,
                                                       double* __restrict__ partials) {
  __shared__ double red[256 * 8];
  const ColMap m(C);
  double* out = partials + (size_t)blockIdx.x * C * 2;
  for (int cg = m.cl; cg < ((m.cv + m.cvp - 1) / m.cvp) * m.cvp; cg += m.cvp) {
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cg < m.cv) {
      for (int row = blockIdx.x * m.rpp + m.rl; row < rows; row += gridDim.x * m.rpp) {
        const float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * ld + 4 * cg);
        const double d0 = a.x, d1 = a.y, d2 = a.z, d3 = a.w;
        v[0] += d0; v[1] += d1; v[2] += d2; v[3] += d3;
        v[4] = fma(d0, d0, v[4]); v[5] = fma(d1, d1, v[5]); v[6] = fma(d2, d2, v[6]); v[7] = fma(d3, d3, v[7]);
      }
    }
    reduce_rows_and_store<8>(m, cg, v, red, out, C);
  }
}

// sums[k] = sum over the R partial rows (k < n), fixed order: 16 row slices per column, combined pairwise.
// (One thread per column walking all R rows was latency-bound: 126 us per call for R = 512.)
__global__ __launch_bounds__(256) void bn_fold_kernel(const double* __restrict__ partials, int R, int n,
                                                      double* __restrict__ sums) {
  __shared__ double red[16][16];
  const int kc = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + kc;
  double t = 0.0;
  if (k < n)
    for (int r = slice; r < R; r += 16) t += partials[(size_t)r * n + k];
  red[slice][kc] = t;
  __syncthreads();
  if (slice == 0 && k < n) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = red[i][kc];
#pragma unroll
    for (int w = 1; w < 16; w <<= 1)
#pragma unroll
      for (int i = 0; i < 16; i += 2 * w) v[i] += v[i + w];
    sums[k] = v[0];
  }
}

// mean, istd from (sum, sum sq, count); running statistics as nn.BatchNorm2d (momentum, unbiased variance)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums /*[C][2] then count*/, int C,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ istd, float* __restrict__ rmean,
                                                          float* __restrict__ rvar) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = sums[2 * C];
  const double mu = sums[2 * c] / n;
  double var = sums[2 * c + 1] / n - mu * mu;
  var = var > 0.0 ? var : 0.0;
  mean[c] = (float)mu;
  istd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
    rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
  }
}

// UP2 variants: x is the block input BEFORE the generator's nearest-neighbour x2 upsample (generator.py:70-82); the output
// pixel (b, h, w) of the (H, W) map reads x at (b, h/2, w/2).  The 4x tensor is never materialised: BatchNorm's statistics of
// an upsampled map are those of the map (every value 4 times), and the backward sums the four children of a source pixel.
__device__ __forceinline__ unsigned up2_src_row(unsigned row, int H, int W) {
  const unsigned w = row % (unsigned)W, t = row / (unsigned)W, h = t % (unsigned)H, b = t / (unsigned)H;
  return (b * (unsigned)(H >> 1) + (h >> 1)) * (unsigned)(W >> 1) + (w >> 1);
}

// y = leaky_relu(((x - mean) * istd) * (1 + gamma) + beta): the normalised activation is never stored
template <bool UP2>
__global__ __launch_bounds__(256) void spade_norm_modulate_fwd_kernel(const float* __restrict__ x, int ld_x,
                                                                      const float* __restrict__ gb, int ld_gb,
                                                                      float* __restrict__ y, int ld_y, int rows, int C,
                                                                      float slope, const float* __restrict__ mean,
                                                                      const float* __restrict__ istd, int H, int W) {
  const int cv = C >> 2;
  const size_t total = (size_t)rows * cv;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / cv;
    const int c = (int)(e - row * cv) * 4;
    const size_t xrow = UP2 ? (size_t)up2_src_row((unsigned)row, H, W) : row;
    const float4 a = *reinterpret_cast<const float4*>(x + xrow * ld_x + c);
    const float4 g = *reinterpret_cast<const float4*>(gb + row * ld_gb + c);
    const float4 b = *reinterpret_cast<const float4*>(gb + row * ld_gb + C + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 is = *reinterpret_cast<const float4*>(istd + c);
    float4 o;
    o.x = lrelu(fmaf((a.x - mu.x) * is.x, 1.f + g.x, b.x), slope);
    o.y = lrelu(fmaf((a.y - mu.y) * is.y, 1.f + g.y, b.y), slope);
    o.z = lrelu(fmaf((a.z - mu.z) * is.z, 1.f + g.z, b.z), slope);
    o.w = lrelu(fmaf((a.w - mu.w) * is.w, 1.f + g.w, b.w), slope);
    *reinterpret_cast<float4*>(y + row * ld_y + c) = o;
  }
}

// backward of the above w.r.t. the NORMALISED activation (dxn) and (gamma | beta), + the per-block partial sums
// (sum dxn, sum dxn*xhat) BatchNorm's backward needs; fixed-column layout so that a thread keeps its channels
// The per-block partial row is (C, 2) (sum dxn, sum dxn*xhat), one unused slot (the count's place in `sums`), then
// (C, 2) (sum dgamma, sum dbeta) -- the column sums of dgb, i.e. the BIAS gradient of the gamma | beta convolution that
// produced gb, which would otherwise re-read dgb (the largest gradient tensor of the step) just to add it up.
// FROMY: the activation's mask is taken from the stored OUTPUT y (rows, C) (its sign is the pre-activation's for slope >= 0)
// and gb holds gamma only (ld_gb = C): the form the fused SphereConv epilogue (eml_sphere_conv_spade_fwd_f32) leaves behind,
// which never stores beta.
template <bool UP2, bool FROMY = false>
__global__ __launch_bounds__(256) void spade_norm_modulate_bwd_kernel(
    const float* __restrict__ gy, int ld_gy, const float* __restrict__ x, int ld_x, const float* __restrict__ gb, int ld_gb,
    float* __restrict__ dxn, int ld_dx, float* __restrict__ dgb, int ld_dgb, int rows, int C, float slope,
    const float* __restrict__ mean, const float* __restrict__ istd, double* __restrict__ partials, int H, int W,
    const float* __restrict__ yout = nullptr) {
  constexpr int NV = 16;
  __shared__ double red[256 * NV];
  const ColMap m(C);
  double* out = partials + (size_t)blockIdx.x * (4 * C + 1);
  if (threadIdx.x == 0) out[2 * C] = 0.0;
  for (int cg = m.cl; cg < ((m.cv + m.cvp - 1) / m.cvp) * m.cvp; cg += m.cvp) {
    double v[NV] = {};
    if (cg < m.cv) {
      const int c = 4 * cg;
      const float4 mu = *reinterpret_cast<const float4*>(mean + c);
      const float4 is = *reinterpret_cast<const float4*>(istd + c);
      for (int row = blockIdx.x * m.rpp + m.rl; row < rows; row += gridDim.x * m.rpp) {
        const float4 u = *reinterpret_cast<const float4*>(gy + (size_t)row * ld_gy + c);
        const size_t xrow = UP2 ? (size_t)up2_src_row((unsigned)row, H, W) : (size_t)row;
        const float4 xv = *reinterpret_cast<const float4*>(x + xrow * ld_x + c);
        const float4 g = *reinterpret_cast<const float4*>(gb + (size_t)row * ld_gb + c);
        // FROMY: `b` is the stored output y of the pixel instead of beta
        const float4 b = FROMY ? *reinterpret_cast<const float4*>(yout + (size_t)row * C + c)
                               : *reinterpret_cast<const float4*>(gb + (size_t)row * ld_gb + C + c);
        const float a[4] = {(xv.x - mu.x) * is.x, (xv.y - mu.y) * is.y, (xv.z - mu.z) * is.z, (xv.w - mu.w) * is.w};
        const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w}, uu[4] = {u.x, u.y, u.z, u.w};
        float d[4], dx[4], dg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d[k] = (FROMY ? bb[k] : fmaf(a[k], 1.f + gg[k], bb[k])) > 0.f ? uu[k] : slope * uu[k];
          dx[k] = d[k] * (1.f + gg[k]);
          dg[k] = d[k] * a[k];
          v[k] += (double)dx[k];
          v[4 + k] = fma((double)dx[k], (double)a[k], v[4 + k]);
          v[8 + k] += (double)dg[k];
          v[12 + k] += (double)d[k];
        }
        *reinterpret_cast<float4*>(dxn + (size_t)row * ld_dx + c) = make_float4(dx[0], dx[1], dx[2], dx[3]);
        *reinterpret_cast<float4*>(dgb + (size_t)row * ld_dgb + c) = make_float4(dg[0], dg[1], dg[2], dg[3]);
        *reinterpret_cast<float4*>(dgb + (size_t)row * ld_dgb + C + c) = make_float4(d[0], d[1], d[2], d[3]);
      }
    }
    reduce_rows_and_store<NV>(m, cg, v, red, out, C);
  }
}

// dx = istd * (dxn - S1/n - xhat * S2/n)   (train mode; sums == NULL: eval mode, dx = istd * dxn).  dx may alias dxn.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dxn, int ld_d, const float* __restrict__ x,
                                                           int ld_x, int rows, int C, const float* __restrict__ mean,
                                                           const float* __restrict__ istd, const double* __restrict__ sums,
                                                           float* __restrict__ dx, int ld_o) {
  const int cv = C >> 2;
  const size_t total = (size_t)rows * cv;
  const double n = sums ? sums[2 * C] : 1.0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / cv;
    const int c = (int)(e - row * cv) * 4;
    const float4 d = *reinterpret_cast<const float4*>(dxn + row * ld_d + c);
    const float4 xv = *reinterpret_cast<const float4*>(x + row * ld_x + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 is = *reinterpret_cast<const float4*>(istd + c);
    const float dd[4] = {d.x, d.y, d.z, d.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
    const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ii[4] = {is.x, is.y, is.z, is.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float m1 = sums ? (float)(sums[2 * (c + k)] / n) : 0.f;
      const float m2 = sums ? (float)(sums[2 * (c + k) + 1] / n) : 0.f;
      o[k] = ii[k] * (dd[k] - m1 - (xx[k] - mm[k]) * ii[k] * m2);
    }
    *reinterpret_cast<float4*>(dx + row * ld_o + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// UP2 backward: dx of the SOURCE pixel = istd * (sum of its 4 children's dxn - 4 S1/n - 4 xhat S2/n), n = the upsampled count
__global__ __launch_bounds__(256) void bn_bwd_apply_up2_kernel(const float* __restrict__ dxn, int ld_d,
                                                               const float* __restrict__ x, int ld_x, int rows_lo, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ istd,
                                                               const double* __restrict__ sums, float* __restrict__ dx,
                                                               int ld_o, int H, int W) {
  const int cv = C >> 2, Hl = H >> 1, Wl = W >> 1;
  const size_t total = (size_t)rows_lo * cv;
  const double n = sums ? sums[2 * C] : 1.0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const unsigned q = (unsigned)(e / cv);
    const int c = (int)(e - (size_t)q * cv) * 4;
    const unsigned wl = q % (unsigned)Wl, t = q / (unsigned)Wl, hl = t % (unsigned)Hl, b = t / (unsigned)Hl;
    const size_t r0 = ((size_t)b * H + 2 * hl) * W + 2 * wl;
    const float4 d0 = *reinterpret_cast<const float4*>(dxn + r0 * ld_d + c);
    const float4 d1 = *reinterpret_cast<const float4*>(dxn + (r0 + 1) * ld_d + c);
    const float4 d2 = *reinterpret_cast<const float4*>(dxn + (r0 + W) * ld_d + c);
    const float4 d3 = *reinterpret_cast<const float4*>(dxn + (r0 + W + 1) * ld_d + c);
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)q * ld_x + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 is = *reinterpret_cast<const float4*>(istd + c);
    const float dd[4] = {(d0.x + d1.x) + (d2.x + d3.x), (d0.y + d1.y) + (d2.y + d3.y), (d0.z + d1.z) + (d2.z + d3.z),
                         (d0.w + d1.w) + (d2.w + d3.w)};
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w}, mm[4] = {mu.x, mu.y, mu.z, mu.w}, ii[4] = {is.x, is.y, is.z, is.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float m1 = sums ? (float)(4.0 * sums[2 * (c + k)] / n) : 0.f;
      const float m2 = sums ? (float)(4.0 * sums[2 * (c + k) + 1] / n) : 0.f;
      o[k] = ii[k] * (dd[k] - m1 - (xx[k] - mm[k]) * ii[k] * m2);
    }
    *reinterpret_cast<float4*>(dx + (size_t)q * ld_o + c) = make_float4(o[0], o[1], o[2], o[3]);
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

// ------------------------------------------------------------------------- parameter-free BatchNorm folded into SPADE
extern "C" int eml_bn_stats_f32(const float* x, int ld, long rows, int C, double* partials, int grid, eml_stream_t stream) {
  if (!x || !partials || rows < 1 || rows > 2147483647L || C < 4 || (C & 3) || bad_ld(ld, C) || grid < 1)
    return eml::fail(EML_EINVAL, "eml_bn_stats_f32: bad arguments");
  hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ld, (int)rows, C, partials);
  return eml::check_launch("eml_bn_stats_f32");
}

extern "C" int eml_bn_fold_f64(const double* partials, int R, int n, double* sums, eml_stream_t stream) {
  if (!partials || !sums || R < 1 || n < 1) return eml::fail(EML_EINVAL, "eml_bn_fold_f64: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, R, n, sums);
  return eml::check_launch("eml_bn_fold_f64");
}

extern "C" int eml_bn_finalize_f32(const double* sums, int C, float eps, float momentum, float* mean, float* istd,
                                   float* running_mean, float* running_var, eml_stream_t stream) {
  if (!sums || !mean || !istd || C < 1 || (running_mean && !running_var) || !(eps > 0.f))
    return eml::fail(EML_EINVAL, "eml_bn_finalize_f32: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, eps, momentum,
                     mean, istd, running_mean, running_var);
  return eml::check_launch("eml_bn_finalize_f32");
}

extern "C" int eml_spade_norm_modulate_fwd_f32(const float* x, int ld_x, const float* gb, int ld_gb, float* y, int ld_y,
                                               long rows, int C, float slope, const float* mean, const float* istd,
                                               eml_stream_t stream) {
  if (!x || !gb || !y || !mean || !istd || rows < 0 || rows > 2147483647L || C < 4 || (C & 3) || bad_ld(ld_x, C) ||
      bad_ld(ld_gb, 2 * C) || bad_ld(ld_y, C))
    return eml::fail(EML_EINVAL, "eml_spade_norm_modulate_fwd_f32: bad arguments");
  if (rows == 0) return EML_OK;
  hipLaunchKernelGGL(spade_norm_modulate_fwd_kernel<false>, dim3(grid_for((size_t)rows * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, ld_x, gb, ld_gb, y, ld_y, (int)rows, C, slope, mean, istd, 0, 0);
  return eml::check_launch("eml_spade_norm_modulate_fwd_f32");
}

extern "C" int eml_bn_bwd_apply_f32(const float* dxn, int ld_d, const float* x, int ld_x, long rows, int C,
                                    const float* mean, const float* istd, const double* sums, float* dx, int ld_o,
                                    eml_stream_t stream) {
  if (!dxn || !x || !mean || !istd || !dx || rows < 0 || rows > 2147483647L || C < 4 || (C & 3) || bad_ld(ld_d, C) ||
      bad_ld(ld_x, C) || bad_ld(ld_o, C))
    return eml::fail(EML_EINVAL, "eml_bn_bwd_apply_f32: bad arguments");
  if (rows == 0) return EML_OK;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, dxn,
                     ld_d, x, ld_x, (int)rows, C, mean, istd, sums, dx, ld_o);
  return eml::check_launch("eml_bn_bwd_apply_f32");
}

// ------------------------------------------------------------ the same three passes with the x2 nearest upsample folded in
namespace {
int up2_args(const char* what, int B, int H, int W, int C) {
  if (B < 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || (C & 3) || (long)B * H * W > 2147483647L)
    return eml::fail(EML_EINVAL, "%s: need even H, W >= 2, C %% 4 == 0 (B=%d, H=%d, W=%d, C=%d)", what, B, H, W, C);
  return EML_OK;
}
}  // namespace

extern "C" int eml_spade_norm_modulate_up2_fwd_f32(const float* x_lo, const float* gb, float* y, int B, int H, int W, int C,
                                                   float slope, const float* mean, const float* istd, eml_stream_t stream) {
  if (!x_lo || !gb || !y || !mean || !istd) return eml::fail(EML_EINVAL, "eml_spade_norm_modulate_up2_fwd_f32: null pointer");
  if (int rc = up2_args("eml_spade_norm_modulate_up2_fwd_f32", B, H, W, C)) return rc;
  const size_t rows = (size_t)B * H * W;
  if (rows == 0) return EML_OK;
  hipLaunchKernelGGL(spade_norm_modulate_fwd_kernel<true>, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     x_lo, C, gb, 2 * C, y, C, (int)rows, C, slope, mean, istd, H, W);
  return eml::check_launch("eml_spade_norm_modulate_up2_fwd_f32");
}

extern "C" int eml_bn_bwd_apply_up2_f32(const float* dxn, const float* x_lo, int B, int H, int W, int C, const float* mean,
                                        const float* istd, const double* sums, float* dx_lo, eml_stream_t stream) {
  if (!dxn || !x_lo || !mean || !istd || !dx_lo) return eml::fail(EML_EINVAL, "eml_bn_bwd_apply_up2_f32: null pointer");
  if (int rc = up2_args("eml_bn_bwd_apply_up2_f32", B, H, W, C)) return rc;
  const size_t rows_lo = (size_t)B * (H / 2) * (W / 2);
  if (rows_lo == 0) return EML_OK;
  hipLaunchKernelGGL(bn_bwd_apply_up2_kernel, dim3(grid_for(rows_lo * (C / 4))), dim3(256), 0, (hipStream_t)stream, dxn, C,
                     x_lo, C, (int)rows_lo, C, mean, istd, sums, dx_lo, C, H, W);
  return eml::check_launch("eml_bn_bwd_apply_up2_f32");
}

// The backward of the modulation that also leaves the column sums of dgb -- the bias gradient of the gamma | beta convolution --
// in its partials: row layout (C, 2) (sum dxn, sum dxn*xhat) | 1 unused slot | (C, 2) (sum dgamma, sum dbeta) = 4C + 1 doubles per
// block, so that eml_bn_fold_f64(partials, grid, 4C + 1, sums) yields `sums` (2C + 1; the caller sets the count at [2C]) followed
// by the (C, 2) table of the bias gradient.  up2 != 0: x_lo is the map before the x2 upsample (H, W = the upsampled size; as
// eml_spade_norm_modulate_up2_bwd_f32); up2 == 0: x (B*H*W, C).  All tensors dense (row stride C, 2C for gb / dgb).
extern "C" int eml_spade_norm_modulate_bwd_cols_f32(const float* gy, const float* x, const float* gb, float* dxn, float* dgb,
                                                    int B, int H, int W, int C, int up2, float slope, const float* mean,
                                                    const float* istd, double* partials, int grid, eml_stream_t stream) {
  if (!gy || !x || !gb || !dxn || !dgb || !mean || !istd || !partials || grid < 1 || B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) ||
      (long)B * H * W > 2147483647L)
    return eml::fail(EML_EINVAL, "eml_spade_norm_modulate_bwd_cols_f32: bad arguments");
  if (up2) {
    if (int rc = up2_args("eml_spade_norm_modulate_bwd_cols_f32", B, H, W, C)) return rc;
    hipLaunchKernelGGL((spade_norm_modulate_bwd_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, gy, C, x, C, gb,
                       2 * C, dxn, C, dgb, 2 * C, B * H * W, C, slope, mean, istd, partials, H, W);
  } else {
    hipLaunchKernelGGL((spade_norm_modulate_bwd_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, gy, C, x, C, gb,
                       2 * C, dxn, C, dgb, 2 * C, B * H * W, C, slope, mean, istd, partials, 0, 0);
  }
  return eml::check_launch("eml_spade_norm_modulate_bwd_cols_f32");
}

// The same pass for the fused SPADE forward (eml_sphere_conv_spade_fwd_f32): `gamma` (B*H*W, C) and the forward's OUTPUT `y`
// (B*H*W, C) instead of the (gamma | beta) tensor -- the activation's mask is y's sign (slope in [0, 1]; slope == 1: y is not
// read for its value, any mask gives dy).  dgb (B*H*W, 2C) = (dgamma | dbeta) as before.
extern "C" int eml_spade_norm_modulate_bwd_y_f32(const float* gy, const float* x, const float* gamma, const float* y, float* dxn,
                                                 float* dgb, int B, int H, int W, int C, int up2, float slope, const float* mean,
                                                 const float* istd, double* partials, int grid, eml_stream_t stream) {
  if (!gy || !x || !gamma || !y || !dxn || !dgb || !mean || !istd || !partials || grid < 1 || B < 1 || H < 1 || W < 1 || C < 4 ||
      (C & 3) || (long)B * H * W > 2147483647L || !(slope >= 0.f && slope <= 1.f))
    return eml::fail(EML_EINVAL, "eml_spade_norm_modulate_bwd_y_f32: bad arguments");
  if (up2) {
    if (int rc = up2_args("eml_spade_norm_modulate_bwd_y_f32", B, H, W, C)) return rc;
    hipLaunchKernelGGL((spade_norm_modulate_bwd_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, gy, C, x, C,
                       gamma, C, dxn, C, dgb, 2 * C, B * H * W, C, slope, mean, istd, partials, H, W, y);
  } else {
    hipLaunchKernelGGL((spade_norm_modulate_bwd_kernel<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, gy, C, x, C,
                       gamma, C, dxn, C, dgb, 2 * C, B * H * W, C, slope, mean, istd, partials, 0, 0, y);
  }
  return eml::check_launch("eml_spade_norm_modulate_bwd_y_f32");
}
