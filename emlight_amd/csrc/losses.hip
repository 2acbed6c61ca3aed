// The generator's L1-type loss terms as ONE launch each way (reference: models/pix2pix_model.py:99-120, loss.py:102-114).
//
//   feature matching  sum_{D i, stage j} mean(|f_ij * m + f_ij * (1 - m) * 50  -  (r_ij * m + r_ij * (1 - m) * 50)|) / num_D
//                     = mean(|(f - r) * (50 - 49 m)|): both sides carry the same per-pixel weight (pix2pix_model.py:101-117)
//   VGG perceptual    sum_i w_i * mean(|vgg_i(fake) - vgg_i(real)|)                                    (loss.py:108-114)
//
// As ATen ops every one of the 8 + 5 terms is sub, mul, abs, mean, div, add forward and sgn, mul, mul, div, a zero-filled
// slice gradient and its copy backward: ~150 launches and a dozen passes over the discriminator's feature maps per step.
// Here a term is a PAIR (f, r, w): n pixel rows of C channels each (channels-last memory), an optional per-row weight w
// (|w| is used: d|x w| = sign(x) |w|), and a scale s (= coefficient / numel).  All pairs of a call travel BY VALUE in the
// kernel arguments (no device table, no host sync):
//   forward   out[0] = sum_pairs s * sum |f - r| |w|          per-workgroup f64 partials, then a fixed-order fold: deterministic
//   backward  g[e] = gout[0] * s * sign(f[e] - r[e]) * |w[row]|, and an optional region gz[0 .. nz) is zero-filled in the same
//             launch (the real half of a discriminator feature map, whose gradient tensor covers fake | real)
#include "eml_common.h"

namespace {

constexpr int kMaxPairs = 16;
constexpr int kGrid = 1024;   // workgroups per pair (a multiple of 64: the fold gives one partial column to each lane)

struct L1Pairs {
  const float* f[kMaxPairs];
  const float* r[kMaxPairs];
  const float* w[kMaxPairs];   // per pixel row, or NULL
  float* g[kMaxPairs];         // backward: gradient w.r.t. f
  float* gz[kMaxPairs];        // backward: region to zero (or NULL)
  long n[kMaxPairs];           // elements = rows * C
  long nz[kMaxPairs];
  int C[kMaxPairs];
  float scale[kMaxPairs];
  int npairs;
};

__device__ __forceinline__ double block_sum_d(double v, double* red /*[4]*/) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void l1_pairs_fwd_kernel(L1Pairs P, double* __restrict__ partial /*[npairs][kGrid]*/) {
  __shared__ double red[4];
  const int q = blockIdx.y;
  const float* __restrict__ f = P.f[q];
  const float* __restrict__ r = P.r[q];
  const float* __restrict__ w = P.w[q];
  const long n = P.n[q];
  const int C = P.C[q];
  double acc = 0.;
  const bool vec = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(r)) & 15) == 0;
  if (vec) {
    const int C4 = C >> 2;
    const long n4 = n >> 2;
    float part = 0.f;   // <= a few dozen terms per thread: f32 here, f64 across threads
#pragma unroll 2
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)n4; i += kGrid * 256u) {   // n < 2^33 (launcher)
      const float4 a = reinterpret_cast<const float4*>(f)[i], b = reinterpret_cast<const float4*>(r)[i];
      const float ww = w ? fabsf(w[i / (unsigned)C4]) : 1.f;
      part += ((fabsf(a.x - b.x) + fabsf(a.y - b.y)) + (fabsf(a.z - b.z) + fabsf(a.w - b.w))) * ww;
    }
    acc = (double)part;
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)kGrid * 256)
      acc += (double)(fabsf(f[i] - r[i]) * (w ? fabsf(w[i / C]) : 1.f));
  }
  const double tot = block_sum_d(acc, red);
  if (threadIdx.x == 0) partial[(size_t)q * kGrid + blockIdx.x] = tot;
}

// one wave per pair: lane l sums partials l, l + 64, ... in order, the wave folds its 64 values with a fixed shuffle tree
__global__ __launch_bounds__(64 * kMaxPairs) void l1_pairs_fold_kernel(L1Pairs P, const double* __restrict__ partial,
                                                                       float* __restrict__ out) {
  __shared__ double term[kMaxPairs];
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.;
  if (q < P.npairs)
    for (int i = lane; i < kGrid; i += 64) s += partial[(size_t)q * kGrid + i];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) term[q] = q < P.npairs ? s * (double)P.scale[q] : 0.;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.;
    for (int i = 0; i < P.npairs; ++i) t += term[i];
    out[0] = (float)t;
  }
}

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void l1_pairs_bwd_kernel(L1Pairs P, const float* __restrict__ gout) {
  const int q = blockIdx.y;
  const float* __restrict__ f = P.f[q];
  const float* __restrict__ r = P.r[q];
  const float* __restrict__ w = P.w[q];
  float* __restrict__ g = P.g[q];
  float* __restrict__ gz = P.gz[q];
  const long n = P.n[q], nz = P.nz[q];
  const int C = P.C[q];
  const float s = gout[0] * P.scale[q];
  const bool vec = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(r) |
                                     reinterpret_cast<uintptr_t>(g)) & 15) == 0;
  if (vec) {
    const int C4 = C >> 2;
    const long n4 = n >> 2;
#pragma unroll 2
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)n4; i += kGrid * 256u) {
      const float4 a = reinterpret_cast<const float4*>(f)[i], b = reinterpret_cast<const float4*>(r)[i];
      const float sw = s * (w ? fabsf(w[i / (unsigned)C4]) : 1.f);
      reinterpret_cast<float4*>(g)[i] = make_float4(sgnf(a.x - b.x) * sw, sgnf(a.y - b.y) * sw, sgnf(a.z - b.z) * sw,
                                                    sgnf(a.w - b.w) * sw);
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)kGrid * 256)
      g[i] = sgnf(f[i] - r[i]) * s * (w ? fabsf(w[i / C]) : 1.f);
  }
  if (gz) {
    if ((nz & 3) == 0 && (reinterpret_cast<uintptr_t>(gz) & 15) == 0) {
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (nz >> 2); i += (long)kGrid * 256)
        reinterpret_cast<float4*>(gz)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nz; i += (long)kGrid * 256) gz[i] = 0.f;
    }
  }
}


// Column sums of a tall, skinny row-major matrix (rows x cols, cols <= 64): the bias gradient of the layers with a handful of
// output channels (conv_img 64 -> 3 at full resolution, the discriminators' heads) -- ATen's sum(0) runs such a shape on a few
// workgroups (0.34 ms for 1 M x 3).  Flat walk with a stride that is a multiple of cols (a thread stays on its column),
// f64 partials per workgroup, fixed-order fold: deterministic.
constexpr int kColGrid = 512;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long n, int cols, long stride,
                                                             double* __restrict__ partial /*[kColGrid][cols]*/) {
  __shared__ double red[256];
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.;
  if (g < stride)
    for (long e = g; e < n; e += stride) acc += (double)x[e];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < cols) {
    // threads of this workgroup on column c: those with (blockIdx.x * 256 + t) % cols == c
    const int first = (int)((threadIdx.x + cols - (int)(((long)blockIdx.x * 256) % cols)) % cols);
    double s = 0.;
    for (int t = first; t < 256; t += cols) s += red[t];
    partial[(size_t)blockIdx.x * cols + threadIdx.x] = s;
  }
}
__global__ __launch_bounds__(64) void colsum_fold_kernel(const double* __restrict__ partial, int nblk, int cols, float* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= cols) return;
  double s = 0.;
  for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * cols + c];
  out[c] = (float)s;
}

int pack(const char* what, int n, const float* const* f, const float* const* r, const float* const* w, const long* rows,
         const int* C, const float* scale, L1Pairs* P) {
  if (n < 1 || n > kMaxPairs) return eml::fail(EML_EINVAL, "%s: 1 <= pairs <= %d (got %d)", what, kMaxPairs, n);
  if (!f || !r || !rows || !C || !scale) return eml::fail(EML_EINVAL, "%s: null argument array", what);
  *P = L1Pairs{};
  P->npairs = n;
  for (int i = 0; i < n; ++i) {
    if (!f[i] || !r[i] || rows[i] < 0 || C[i] < 1) return eml::fail(EML_EINVAL, "%s: pair %d: null pointer or empty shape", what, i);
    if (rows[i] > (1L << 33) / C[i]) return eml::fail(EML_EINVAL, "%s: pair %d: more than 2^33 elements", what, i);
    P->f[i] = f[i];
    P->r[i] = r[i];
    P->w[i] = w ? w[i] : nullptr;
    P->n[i] = rows[i] * C[i];
    P->C[i] = C[i];
    P->scale[i] = scale[i];
  }
  return EML_OK;
}

}  // namespace

extern "C" size_t eml_l1_pairs_partial_doubles(int npairs) { return npairs > 0 ? (size_t)npairs * kGrid : 0; }

extern "C" int eml_l1_pairs_fwd_f32(int npairs, const float* const* f, const float* const* r, const float* const* w,
                                    const long* rows, const int* C, const float* scale, double* partial, float* out,
                                    eml_stream_t stream) {
  L1Pairs P;
  if (int e = pack("eml_l1_pairs_fwd_f32", npairs, f, r, w, rows, C, scale, &P)) return e;
  if (!partial || !out) return eml::fail(EML_EINVAL, "eml_l1_pairs_fwd_f32: null partial / out");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_pairs_fwd_kernel, dim3(kGrid, npairs), dim3(256), 0, st, P, partial);
  hipLaunchKernelGGL(l1_pairs_fold_kernel, dim3(1), dim3(64 * kMaxPairs), 0, st, P, partial, out);
  return eml::check_launch("eml_l1_pairs_fwd_f32");
}

extern "C" int eml_l1_pairs_bwd_f32(int npairs, const float* const* f, const float* const* r, const float* const* w,
                                    const long* rows, const int* C, const float* scale, const float* gout,
                                    float* const* g, float* const* gzero, const long* nzero, eml_stream_t stream) {
  L1Pairs P;
  if (int e = pack("eml_l1_pairs_bwd_f32", npairs, f, r, w, rows, C, scale, &P)) return e;
  if (!gout || !g) return eml::fail(EML_EINVAL, "eml_l1_pairs_bwd_f32: null gout / g");
  for (int i = 0; i < npairs; ++i) {
    if (!g[i]) return eml::fail(EML_EINVAL, "eml_l1_pairs_bwd_f32: pair %d: null gradient pointer", i);
    P.g[i] = g[i];
    P.gz[i] = gzero ? gzero[i] : nullptr;
    P.nz[i] = (gzero && gzero[i] && nzero) ? nzero[i] : 0;
    if (P.nz[i] < 0) return eml::fail(EML_EINVAL, "eml_l1_pairs_bwd_f32: pair %d: negative zero-fill length", i);
  }
  hipLaunchKernelGGL(l1_pairs_bwd_kernel, dim3(kGrid, npairs), dim3(256), 0, (hipStream_t)stream, P, gout);
  return eml::check_launch("eml_l1_pairs_bwd_f32");
}

extern "C" size_t eml_colsum_partial_doubles(int cols) { return cols > 0 ? (size_t)kColGrid * cols : 0; }

extern "C" int eml_colsum_f32(const float* x, long rows, int cols, double* partial, float* out, eml_stream_t stream) {
  if (!x || !partial || !out || rows < 0 || cols < 1 || cols > 64)
    return eml::fail(EML_EINVAL, "eml_colsum_f32: null pointer, negative rows or cols outside [1, 64] (cols=%d)", cols);
  const long n = rows * cols;
  // every workgroup takes part (its partial row must be written); the stride is the largest multiple of cols <= the grid
  const long threads = (long)kColGrid * 256;
  const long stride = threads / cols * cols;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(kColGrid), dim3(256), 0, st, x, n, cols, stride, partial);
  hipLaunchKernelGGL(colsum_fold_kernel, dim3(1), dim3(64), 0, st, partial, kColGrid, cols, out);
  return eml::check_launch("eml_colsum_f32");
}
