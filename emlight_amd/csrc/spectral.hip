// Spectral normalisation of a 3x3 convolution weight, fused with the re-layout the gather-GEMM kernels want
// (reference: models/networks/normalization.py:24-33 'spectral' -> torch.nn.utils.spectral_norm on every SphereConv2D of the
// generator's ResNet blocks and of the discriminator; architecture.py:41-45).
//
// torch's hook runs, per wrapped convolution and per forward: two gemv, two norms, two divisions, a dot, the division of
// the weight by sigma, clones of u and v -- and the SphereConv then copies the result into its (O, tap, c) operand layout:
// a dozen launches of 4-10 us for 29 convolutions, twice per iteration, and as many again in the backward.  Here:
//   forward   t = W^T u (partials over row slices) -> v = t / max(|t|, eps) -> s = W v -> u = s / max(|s|, eps),
//             sigma = u . s, W2[o][tap*C + c] = W[o][c*9 + tap] / sigma                              (5 launches)
//   backward  dW[o][c*9 + tap] = (dW2[o][tap*C + c] - <dW2, W2> u[o] v[c*9 + tap]) / sigma          (2 launches)
// W (O, K = 9C) is the parameter in its natural (o, c, kh, kw) order; u (O), v (K) are the module's buffers, updated in place
// when `iterate` (training mode, one power iteration as the reference's default), read-only otherwise.  u, v are constants
// of the backward (torch detaches them): d sigma / dW = u v^T.
#include <algorithm>

#include "eml_common.h"

namespace {

constexpr int kMaxSlices = 16;

__device__ __forceinline__ double block_sum(double v, double* red /*[16]*/) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// The five forward phases are written as device functions of (operands, block coordinates): the launches below run them for a
// BATCH of weights at once (SnBatch), each workgroup looking up its weight and its block within it.

// t_part[slice][k] = sum_{o in slice} W[o][k] * u[o]
__device__ __forceinline__ void sn_wt_u(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ t_part, int O,
                                        int K, int rows_per_slice, int bx, int by) {
  const int k = bx * 256 + threadIdx.x;
  if (k >= K) return;
  const int o0 = by * rows_per_slice, o1 = min(O, o0 + rows_per_slice);
  float a0 = 0.f, a1 = 0.f;
  int o = o0;
  for (; o + 1 < o1; o += 2) {
    a0 = fmaf(W[(size_t)o * K + k], u[o], a0);
    a1 = fmaf(W[(size_t)(o + 1) * K + k], u[o + 1], a1);
  }
  if (o < o1) a0 = fmaf(W[(size_t)o * K + k], u[o], a0);
  t_part[(size_t)by * K + k] = a0 + a1;
}

// t = sum of the slices, and this workgroup's share of |t|^2 (the norm is finished by its consumers: 36 values at most)
__device__ __forceinline__ void sn_fold_t(const float* __restrict__ t_part, int S, int K, float* __restrict__ t,
                                          double* __restrict__ tnorm_part, int bx, double* red) {
  const int k = bx * 256 + threadIdx.x;
  float a = 0.f;
  if (k < K) {
    for (int s = 0; s < S; ++s) a += t_part[(size_t)s * K + k];
    t[k] = a;
  }
  const double tot = block_sum((double)a * a, red);
  if (threadIdx.x == 0) tnorm_part[bx] = tot;
}

__device__ __forceinline__ float inv_norm(const double* __restrict__ part, int n, float eps) {
  double q = 0.;
  for (int i = 0; i < n; ++i) q += part[i];   // fixed order: every caller gets the same bits
  return 1.f / fmaxf((float)sqrt(q), eps);
}

// s[o] = sum_k W[o][k] v[k], v = t * vscale (vscale = 1 / max(|t|, eps) from the partials when tnorm_part != NULL): one wave per row
__device__ __forceinline__ void sn_w_v(const float* __restrict__ W, const float* __restrict__ v, const double* __restrict__ tnorm_part,
                                       int np, float eps, float* __restrict__ s, int O, int K, int bx) {
  const int o = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  const float vscale = tnorm_part ? inv_norm(tnorm_part, np, eps) : 1.f;
  const float* row = W + (size_t)o * K;
  float a = 0.f;
  if ((K & 3) == 0) {
    for (int k = 4 * lane; k < K; k += 256) {
      const float4 w = *reinterpret_cast<const float4*>(row + k), x = *reinterpret_cast<const float4*>(v + k);
      a = fmaf(w.x, x.x, a); a = fmaf(w.y, x.y, a); a = fmaf(w.z, x.z, a); a = fmaf(w.w, x.w, a);
    }
  } else {
    for (int k = lane; k < K; k += 64) a = fmaf(row[k], v[k], a);
  }
  a = eml::wave_sum(a) * vscale;
  if (lane == 0) s[o] = a;
}

// sigma, u and v of this forward, by ONE workgroup (the O-vector s is at most a few KB): sigma = u . (W v), u = s / max(|s|, eps)
// when `iterate`, the stored u otherwise; the (u | v) actually used are kept for the backward.
__device__ __forceinline__ void sn_sigma(const float* __restrict__ s, float* __restrict__ u, float* __restrict__ v,
                                         const float* __restrict__ t, const double* __restrict__ tnorm_part, int np, int iterate,
                                         float eps, float* __restrict__ sigma_out, float* __restrict__ u_used, int O, int K,
                                         double* red) {
  double q = 0.;
  for (int i = threadIdx.x; i < O; i += 256) q += iterate ? (double)s[i] * s[i] : (double)s[i] * u[i];
  const double tot = block_sum(q, red);
  float sigma, unorm = 1.f;
  if (iterate) {
    unorm = 1.f / fmaxf((float)sqrt(tot), eps);       // u = s / max(|s|, eps)
    sigma = (float)(tot * (double)unorm);             // u . (W v) = |s|^2 / max(|s|, eps)
  } else {
    sigma = (float)tot;
  }
  for (int i = threadIdx.x; i < O; i += 256) {
    const float ui = iterate ? s[i] * unorm : u[i];
    if (iterate) u[i] = ui;
    u_used[i] = ui;
  }
  const float vscale = iterate ? inv_norm(tnorm_part, np, eps) : 1.f;
  for (int i = threadIdx.x; i < K; i += 256) {   // the backward's constants: (u | v) as used here
    const float vi = iterate ? t[i] * vscale : v[i];
    if (iterate) v[i] = vi;
    u_used[O + i] = vi;
  }
  if (threadIdx.x == 0) *sigma_out = sigma;
}

// The re-layout itself is a pure stream (read W once, write W2 once: 2 x 289 MB per generator pass at ngf = 64): one
// workgroup per (row o, chunk of kCC channels), 16-byte accesses on both sides, the (c, tap) -> (tap, c) transpose of the
// chunk through LDS.  LDS index of element (c, tap) = 9c + tap + (c >> 2): a lane's four channels of one tap sit 9 apart, the
// lanes of a wave 37 apart -- odd, so neither side conflicts.  (Round 4's kernel gave a whole row of up to 9216 floats to
// one workgroup, 4-byte accesses, one row in flight per workgroup: 0.18 TB/s, 6 ms of a joint step.)
constexpr int kCC = 256;
constexpr int kTile = kCC * 9 + kCC / 4;

// one chunk: dst[tap*C + c] = src[c*9 + tap] * inv for the cc channels of the chunk (src / dst already point at the chunk)
__device__ __forceinline__ void relayout_chunk(const float* __restrict__ src, float* __restrict__ dst, int C, int cc, float inv,
                                               bool vec, float* tile) {
  const int tid = threadIdx.x;
  if (vec) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    for (int j = tid; j < (cc * 9) / 4; j += 256) {       // elements 4j .. 4j+3 of the chunk: 36 | 4j' boundaries only
      const float4 w = src4[j];
      const int b = 4 * j + j / 9;
      tile[b] = w.x; tile[b + 1] = w.y; tile[b + 2] = w.z; tile[b + 3] = w.w;
    }
    __syncthreads();
    const int c4 = tid & 63, q = cc >> 2;
    if (c4 < q) {
      for (int tap = tid >> 6; tap < 9; tap += 4) {
        const float* r = tile + 37 * c4 + tap;
        *reinterpret_cast<float4*>(dst + (size_t)tap * C + 4 * c4) = make_float4(r[0] * inv, r[9] * inv, r[18] * inv, r[27] * inv);
      }
    }
  } else {
    for (int i = tid; i < cc * 9; i += 256) tile[i + i / 36] = src[i];
    __syncthreads();
    for (int j = tid; j < cc * 9; j += 256) {
      const int tap = j / cc, c = j - tap * cc;
      dst[(size_t)tap * C + c] = tile[9 * c + tap + (c >> 2)] * inv;
    }
  }
}

// ---- the batch: up to kMaxItems weights per launch, passed by value (2.3 KB of kernel arguments)
constexpr int kMaxItems = 32;
struct SnItem {
  const float* W;   // (O, 9C): the parameter
  float *u, *v;     // the module's buffers
  float *W2, *sigma, *uv, *scratch;
  int O, C;
};
struct SnBatch {
  SnItem it[kMaxItems];
  int n, iterate;
  float eps;
};
__host__ __device__ inline int sn_np(int K) { return (K + 255) / 256; }
__host__ __device__ inline int sn_slices(int O) { return O / 32 < 1 ? 1 : (O / 32 > kMaxSlices ? kMaxSlices : O / 32); }
// workgroups of one weight in phase PH: 0 W^T u partials, 1 fold, 2 W v, 3 sigma, 4 re-layout
template <int PH>
__host__ __device__ inline int sn_blocks(int O, int C) {
  const int K = 9 * C;
  return PH == 0 ? sn_np(K) * sn_slices(O) : PH == 1 ? sn_np(K) : PH == 2 ? (O + 3) / 4 : PH == 3 ? 1 : O * ((C + kCC - 1) / kCC);
}
// scratch of one weight: t partials | t | s | (8-byte aligned) |t|^2 partials
struct SnScratch {
  float *t_part, *t, *s;
  double* tnorm_part;
};
__host__ __device__ inline SnScratch sn_scratch(float* scratch, int O, int K) {
  SnScratch r;
  r.t_part = scratch;
  r.t = scratch + (size_t)kMaxSlices * K;
  r.s = r.t + K;
  r.tnorm_part = reinterpret_cast<double*>(scratch + (((size_t)(kMaxSlices + 1) * K + O + 1) & ~(size_t)1));
  return r;
}

template <int PH>
__global__ __launch_bounds__(256) void sn_phase_kernel(SnBatch b) {
  __shared__ double red[16];
  __shared__ float tile[PH == 4 ? kTile : 1];
  // which weight, which of its blocks (a scalar walk over at most 32 entries of the kernel arguments)
  int item = 0, local = (int)blockIdx.x;
  for (; item < b.n; ++item) {
    const int cnt = sn_blocks<PH>(b.it[item].O, b.it[item].C);
    if (local < cnt) break;
    local -= cnt;
  }
  if (item >= b.n) return;
  const SnItem& w = b.it[item];
  const int O = w.O, C = w.C, K = 9 * C, np = sn_np(K);
  const SnScratch sc = sn_scratch(w.scratch, O, K);
  if constexpr (PH == 0) {
    const int slices = sn_slices(O), rps = (O + slices - 1) / slices;
    sn_wt_u(w.W, w.u, sc.t_part, O, K, rps, local % np, local / np);
  } else if constexpr (PH == 1) {
    sn_fold_t(sc.t_part, sn_slices(O), K, sc.t, sc.tnorm_part, local, red);
  } else if constexpr (PH == 2) {
    if (b.iterate) sn_w_v(w.W, sc.t, sc.tnorm_part, np, b.eps, sc.s, O, K, local);
    else sn_w_v(w.W, w.v, nullptr, 0, b.eps, sc.s, O, K, local);
  } else if constexpr (PH == 3) {
    sn_sigma(sc.s, w.u, w.v, sc.t, sc.tnorm_part, np, b.iterate, b.eps, w.sigma, w.uv, O, K, red);
  } else {
    // W2[o][tap*C + c] = W[o][c*9 + tap] / sigma
    const int o = local % O, c0 = (local / O) * kCC, cc = min(kCC, C - c0);
    const bool vec = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(w.W) | reinterpret_cast<uintptr_t>(w.W2)) & 15) == 0;
    relayout_chunk(w.W + (size_t)o * K + (size_t)c0 * 9, w.W2 + (size_t)o * K + c0, C, cc, 1.f / *w.sigma, vec, tile);
  }
}

// SPADE's two heads (normalization.py:96-98: mlp_gamma, mlp_beta, each (Cn, C, 3, 3) + bias) as ONE (2 Cn, 9 C) operand in the
// (tap, c) column order of the gather-GEMM kernels -- one launch instead of cat(weights), cat(biases) and the re-layout copy.
// reorder != 0: rows in the order of eml_sphere_conv_spade_fwd_f32 (position p holds head `half`, channel c with
// c = 64 (p / 128) + 32 ((p % 128) / 64) + p % 32, half = (p % 64) / 32); else cat order (gamma rows, then beta rows).
__global__ __launch_bounds__(256) void spade_heads_kernel(const float* __restrict__ Wg, const float* __restrict__ Wb,
                                                          const float* __restrict__ bg, const float* __restrict__ bb,
                                                          float* __restrict__ W2, float* __restrict__ b2, int Cn, int C,
                                                          int reorder) {
  __shared__ float tile[kTile];
  const int K = 9 * C, p = blockIdx.x, c0 = blockIdx.y * kCC, cc = min(kCC, C - c0);
  int half, c;
  if (reorder) {
    c = 64 * (p / 128) + 32 * ((p % 128) / 64) + p % 32;
    half = (p % 64) / 32;
  } else {
    half = p >= Cn;
    c = p - half * Cn;
  }
  const float* W = half ? Wb : Wg;
  if (b2 && blockIdx.y == 0 && threadIdx.x == 0) b2[p] = (half ? bb : bg)[c];
  const bool vec = (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(Wg) | reinterpret_cast<uintptr_t>(Wb) |
                                     reinterpret_cast<uintptr_t>(W2)) & 15) == 0;
  relayout_chunk(W + (size_t)c * K + (size_t)c0 * 9, W2 + (size_t)p * K + c0, C, cc, 1.f, vec, tile);
}

// partial[b] = sum over this workgroup's elements of dW2 * W2
__global__ __launch_bounds__(256) void sn_inner_kernel(const float* __restrict__ dW2, const float* __restrict__ W2, size_t n,
                                                       double* __restrict__ partial) {
  __shared__ double red[16];
  double q = 0.;
  const size_t n4 = (((reinterpret_cast<uintptr_t>(dW2) | reinterpret_cast<uintptr_t>(W2)) & 15) == 0) ? n >> 2 : 0;
  const float4* a4 = reinterpret_cast<const float4*>(dW2);
  const float4* b4 = reinterpret_cast<const float4*>(W2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 a = a4[i], b = b4[i];
    q += ((double)a.x * b.x + (double)a.y * b.y) + ((double)a.z * b.z + (double)a.w * b.w);
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q += (double)dW2[i] * W2[i];
  const double tot = block_sum(q, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// dW[o][c*9 + tap] = (dW2[o][tap*C + c] - <dW2, W2> u[o] v[c*9 + tap]) / sigma: the transposed twin of sn_relayout_kernel
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ dW2, const double* __restrict__ partial,
                                                           int P, const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ sigma, float* __restrict__ dW, int C) {
  __shared__ float tile[kTile];
  const int K = 9 * C, o = blockIdx.x, c0 = blockIdx.y * kCC, cc = min(kCC, C - c0), tid = threadIdx.x;
  double inner = 0.;
  for (int i = 0; i < P; ++i) inner += partial[i];   // fixed order, same in every workgroup
  const float inv = 1.f / *sigma;
  const float coef = (float)inner * u[o];
  const float* src = dW2 + (size_t)o * K + c0;
  float* dst = dW + (size_t)o * K + (size_t)c0 * 9;
  const float* vv = v + (size_t)c0 * 9;
  if ((C & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(dW2) | reinterpret_cast<uintptr_t>(dW) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
    const int c4 = tid & 63, q = cc >> 2;
    if (c4 < q) {
      for (int tap = tid >> 6; tap < 9; tap += 4) {
        const float4 g = *reinterpret_cast<const float4*>(src + (size_t)tap * C + 4 * c4);
        float* r = tile + 37 * c4 + tap;
        r[0] = g.x; r[9] = g.y; r[18] = g.z; r[27] = g.w;
      }
    }
    __syncthreads();
    for (int j = tid; j < (cc * 9) / 4; j += 256) {
      const int b = 4 * j + j / 9;
      const float4 x = *reinterpret_cast<const float4*>(vv + 4 * j);
      *reinterpret_cast<float4*>(dst + 4 * j) = make_float4((tile[b] - coef * x.x) * inv, (tile[b + 1] - coef * x.y) * inv,
                                                           (tile[b + 2] - coef * x.z) * inv, (tile[b + 3] - coef * x.w) * inv);
    }
  } else {
    for (int j = tid; j < cc * 9; j += 256) {
      const int tap = j / cc, c = j - tap * cc;
      tile[9 * c + tap + (c >> 2)] = src[(size_t)tap * C + c];
    }
    __syncthreads();
    for (int i = tid; i < cc * 9; i += 256) dst[i] = (tile[i + i / 36] - coef * vv[i]) * inv;
  }
}

constexpr int kInnerGrid = 256;

}  // namespace

extern "C" size_t eml_spectral_norm_scratch_floats(int O, int C) {
  if (O < 1 || C < 1) return 0;
  return (size_t)(kMaxSlices + 1) * 9 * C + O + 2 * ((9 * C + 255) / 256 + 1);   // t partials, t, s, |t|^2 partials (doubles)
}

namespace {
template <int PH>
void sn_launch_phase(const SnBatch& b, hipStream_t st) {
  long blocks = 0;
  for (int i = 0; i < b.n; ++i) blocks += sn_blocks<PH>(b.it[i].O, b.it[i].C);
  hipLaunchKernelGGL(sn_phase_kernel<PH>, dim3((unsigned)blocks), dim3(256), 0, st, b);
}
void sn_launch(const SnBatch& b, hipStream_t st) {
  if (b.iterate) {
    sn_launch_phase<0>(b, st);
    sn_launch_phase<1>(b, st);
  }
  sn_launch_phase<2>(b, st);
  sn_launch_phase<3>(b, st);
  sn_launch_phase<4>(b, st);
}
const char* sn_check_item(const float* W, const float* u, const float* v, const float* W2, const float* sigma, const float* uv,
                          const float* scratch, int O, int C) {
  if (!W || !u || !v || !W2 || !sigma || !uv || !scratch || O < 1 || C < 1) return "null pointer or empty shape";
  if ((size_t)9 * C * sizeof(float) > 160 * 1024) return "C too wide";
  return nullptr;
}
}  // namespace

extern "C" int eml_spectral_norm_w2_f32(const float* W, float* u, float* v, int iterate, float eps, float* W2, float* sigma,
                                        float* uv_used /*[O + 9C]*/, float* scratch, int O, int C, eml_stream_t stream) {
  if (const char* why = sn_check_item(W, u, v, W2, sigma, uv_used, scratch, O, C))
    return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_f32: %s (O=%d, C=%d)", why, O, C);
  if (!(eps > 0.f)) return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_f32: eps must be positive");
  SnBatch b;
  b.it[0] = SnItem{W, u, v, W2, sigma, uv_used, scratch, O, C};
  b.n = 1;
  b.iterate = iterate ? 1 : 0;
  b.eps = eps;
  sn_launch(b, (hipStream_t)stream);
  return eml::check_launch("eml_spectral_norm_w2_f32");
}

// The same for n weights in 5 launches altogether (ceil(n / 32) x 5): every weight of a network that is spectrally normalised
// at the top of its forward, instead of 5 launches of 5-10 us in front of each convolution.  All weights share `iterate` and
// `eps`; item i uses its own buffers exactly as eml_spectral_norm_w2_f32 does (the results are bit-identical to n such calls).
extern "C" int eml_spectral_norm_w2_batch_f32(int n, const float* const* W, float* const* u, float* const* v, int iterate, float eps,
                                              float* const* W2, float* const* sigma, float* const* uv_used,
                                              float* const* scratch, const int* O, const int* C, eml_stream_t stream) {
  if (n < 0 || (n > 0 && (!W || !u || !v || !W2 || !sigma || !uv_used || !scratch || !O || !C)))
    return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_batch_f32: null array or negative count");
  if (!(eps > 0.f)) return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_batch_f32: eps must be positive");
  for (int i = 0; i < n; ++i)
    if (const char* why = sn_check_item(W[i], u[i], v[i], W2[i], sigma[i], uv_used[i], scratch[i], O[i], C[i]))
      return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_batch_f32: item %d: %s (O=%d, C=%d)", i, why, O[i], C[i]);
  for (int i0 = 0; i0 < n; i0 += kMaxItems) {
    SnBatch b;
    b.n = std::min(kMaxItems, n - i0);
    b.iterate = iterate ? 1 : 0;
    b.eps = eps;
    for (int j = 0; j < b.n; ++j) {
      const int i = i0 + j;
      b.it[j] = SnItem{W[i], u[i], v[i], W2[i], sigma[i], uv_used[i], scratch[i], O[i], C[i]};
    }
    sn_launch(b, (hipStream_t)stream);
  }
  return eml::check_launch("eml_spectral_norm_w2_batch_f32");
}

extern "C" int eml_spectral_norm_w2_bwd_f32(const float* dW2, const float* W2, const float* u_used, const float* v,
                                            const float* sigma, double* partial /*[256]*/, float* dW, int O, int C,
                                            eml_stream_t stream) {
  if (!dW2 || !W2 || !u_used || !v || !sigma || !partial || !dW || O < 1 || C < 1)
    return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_bwd_f32: null pointer or empty shape");
  const int K = 9 * C;
  if ((size_t)K * sizeof(float) > 160 * 1024) return eml::fail(EML_EINVAL, "eml_spectral_norm_w2_bwd_f32: C = %d too wide", C);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)O * K;
  const int grid = (int)std::min<size_t>(kInnerGrid, (n + 4095) / 4096);
  hipLaunchKernelGGL(sn_inner_kernel, dim3(grid), dim3(256), 0, st, dW2, W2, n, partial);
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(O, (C + kCC - 1) / kCC), dim3(256), 0, st, dW2, partial, grid, u_used, v, sigma, dW,
                     C);
  return eml::check_launch("eml_spectral_norm_w2_bwd_f32");
}

extern "C" int eml_spade_heads_w2_f32(const float* Wg, const float* Wb, const float* bg, const float* bb, float* W2, float* b2,
                                      int Cn, int C, int reorder, eml_stream_t stream) {
  if (!Wg || !Wb || !W2 || Cn < 1 || C < 1) return eml::fail(EML_EINVAL, "eml_spade_heads_w2_f32: null pointer or empty shape");
  if (b2 && (!bg || !bb)) return eml::fail(EML_EINVAL, "eml_spade_heads_w2_f32: b2 needs both head biases");
  if (reorder && Cn % 64) return eml::fail(EML_EINVAL, "eml_spade_heads_w2_f32: the one-launch SPADE's row order needs Cn %% 64 == 0 (Cn=%d)", Cn);
  hipLaunchKernelGGL(spade_heads_kernel, dim3(2 * Cn, (C + kCC - 1) / kCC), dim3(256), 0, (hipStream_t)stream, Wg, Wb, bg, bb, W2,
                     b2, Cn, C, reorder ? 1 : 0);
  return eml::check_launch("eml_spade_heads_w2_f32");
}
