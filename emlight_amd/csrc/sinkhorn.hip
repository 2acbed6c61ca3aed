// EMLight's "spherical mover's" loss: debiased Sinkhorn divergence over N sphere anchors.
//
// Replaces the whole of SamplesLoss.sinkhorn_tensorized (reference
// RegressionNetwork/geomloss/samples_loss.py:79-92 + sinkhorn_divergence.py:72-109): there a
// step costs 4 materialised (B,N,N) cost tensors and ~200-300 tiny ATen launches
// (4*(n_eps+2) softmins x ~6 ops) plus a .item() host sync for the diameter.
//
// Here one C-ABI call enqueues the loop kernel (which also derives the diameter and the epsilon
// schedule on the device, redundantly per workgroup) + a tiny finishing kernel.  The four coupled
// softmin problems      g=0 xx -> a_x   g=1 yy -> b_y   g=2 yx -> a_y   g=3 xy -> b_x
// each own 8 wavefronts (512 threads); a sample is two 1024-thread workgroups -- role 0 runs
// the independent pair (xx, yy), role 1 the coupled pair (yx, xy) -- so 2*B workgroups spread
// over the CUs at 4 waves per SIMD.
//
// N <= 128 ("cached" kernel): four lanes share row i; thread (i, quarter) keeps its <=32 costs
// C_ij = .5*(.1*(p_i-q_j)^2 + M_ij) in REGISTERS for the whole eps-scaling loop
// (the cost matrices never exist in HBM; M is staged once through LDS with coalesced
// reads).  A sweep is one fma+max pass and one fma+exp2+add pass over those registers, the
// row reduction is two lane shuffles, and the dual vectors h = log w + f/eps
// travel between the two groups through a double-buffered LDS array with ONE barrier per
// sweep.  128 < N <= 512, N % 4 == 0 ("tiled" kernel): the chord matrix streams through LDS in column tiles shared by
// both problems of the workgroup, online log-sum-exp per row.  Other N ("stream" kernel): thread = row, costs
// recomputed per sweep from Mt (coalesced, L2-resident).
//
// Roofline: algorithmic bytes per eps-step = 4*(B*N^2*4 + 2*B*N*4), i.e. the
// reference's materialised-cost traffic; this kernel's real HBM traffic is x, y, M and the
// outputs only, so it is latency/transcendental-bound: (n_eps+2)*4*N^2 exp2 per sample.
#include "eml_common.h"

#include <cmath>

// ---- phase timestamps of the cached kernel, experiment builds only (tools/exp_build.sh stamps -DEML_STAMPS, then
// EML_LIB_PATH=build_exp/lib_stamps.so python tools/sinkhorn_stamps.py): thread 0 of workgroup 1 (sample 0's coupled pair)
// stores the 100 MHz wall clock at the phase boundaries.  This is how the prologue's share was found;
// in the product build EML_STAMP compiles to nothing and the symbol below does not exist.
#ifdef EML_STAMPS
__device__ long long eml_sinkhorn_stamp_buf[64];
extern "C" int eml_sinkhorn_read_stamps(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eml_sinkhorn_stamp_buf), sizeof(long long) * 64);
}
#define EML_STAMP(k)                                                                            \
  do {                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x == 1 && (k) < 64) eml_sinkhorn_stamp_buf[k] = wall_clock64(); \
  } while (0)
#else
#define EML_STAMP(k) \
  do {               \
  } while (0)
#endif

namespace {

constexpr int kJPT = 64;   // LDS padding unit (unrolled reads past a row's end stay in-bounds)
constexpr int kCJ = 32;    // cached kernel: costs per thread; 4 lanes share a row (N <= 4*kCJ = 128)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kBig = 1e30f;

__host__ __device__ __forceinline__ int round_up4(int v) { return (v + 3) & ~3; }

// C_ij following geomloss/utils.py:90-94 and the /2 of samples_loss.py:82
__device__ __forceinline__ float cost_ij(float p, float q, float m) {
  const float d = (p * p - 2.0f * (p * q) + q * q) * 0.1f + m;
  return d * 0.5f;
}

// LDS carve-up (floats).  NP = round_up4(N) + kJPT so unrolled reads past a row's end stay
// in-bounds; everything is 16-B aligned.
//   pts [2][NP]  x, y            lw2 [2][NP]  log2(e)*log-weights (alpha, beta)
//   h2  [2][2][NP] double-buffered log2(e)*(log w + f/eps), per consuming local group
//   pot [2][NP]  potentials (stream kernel)      M [N][ldm] (cached kernel)
constexpr int kSmemVecs = 2 + 2 + 4 + 2;

// ---- epsilon schedule, computed by every workgroup (no separate launch, no host sync):
// d = diameter > 0 ? diameter : range(x U y) over the WHOLE batch (sinkhorn_divergence.py:9-18),
// eps_s = [d^p] + [exp(e) for e in arange(p ln d, p ln blur, p ln scaling)] + [blur^p] in f64 like numpy
// The scan's first trip (four float4 per thread) is a separate step so that a kernel can REQUEST it early and do other
// work (staging) while it is in flight: schedule_scan_begin issues the loads, device_schedule folds them.
struct ScanHead {
  float4 a, c, a2, c2;
  float rlo, rhi;   // range of x U y on the OTHER ranks (all-reduced by the caller), or +inf / -inf
};
template <int kWG>
__device__ __forceinline__ ScanHead schedule_scan_begin(const float* __restrict__ x, const float* __restrict__ y, int B,
                                                        int N, double diameter, const float* __restrict__ range_dev) {
  ScanHead h;
  h.a = h.c = h.a2 = h.c2 = make_float4(0.f, 0.f, 0.f, 0.f);
  h.rlo = INFINITY;
  h.rhi = -INFINITY;
  if (diameter <= 0.0 && range_dev) {   // data-parallel ranks: the diameter is the range over the GLOBAL batch
    h.rlo = range_dev[0];
    h.rhi = range_dev[1];
  }
  const long n4 = ((long)B * N) >> 2;   // hipMalloc'd buffers: 16-B aligned
  if (diameter <= 0.0 && n4 > 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const long k = min((long)threadIdx.x, n4 - 1), k2 = min((long)threadIdx.x + kWG, n4 - 1);   // clamped: min / max idempotent
    h.a = x4[k];
    h.c = y4[k];
    h.a2 = x4[k2];
    h.c2 = y4[k2];
  }
  return h;
}

template <int kWG, bool kFinalBarrier = true>
__device__ __forceinline__ void device_schedule(const ScanHead& head, const float* __restrict__ x,
                                                const float* __restrict__ y, int B, int N,
                                                double blur, double log_blur, double log_scaling, int p_exp,
                                                double diameter, float* eps_l,
                                                int* n_eps_l, float* __restrict__ eps_out, int* __restrict__ n_eps_out,
                                                float* __restrict__ diameter_out) {
  __shared__ float red_lo[16], red_hi[16];
  const int tid0 = threadIdx.x;
  float lo = INFINITY, hi = -INFINITY;
  if (diameter <= 0.0) {
    const long n_all = (long)B * N, n4 = n_all >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    auto fold = [&](const float4& a, const float4& c2) {
      lo = fminf(fminf(fminf(lo, fminf(a.x, a.y)), fminf(a.z, a.w)), fminf(fminf(c2.x, c2.y), fminf(c2.z, c2.w)));
      hi = fmaxf(fmaxf(fmaxf(hi, fmaxf(a.x, a.y)), fmaxf(a.z, a.w)), fmaxf(fmaxf(c2.x, c2.y), fmaxf(c2.z, c2.w)));
    };
    if (n4 > 0) {
      fold(head.a, head.c);
      fold(head.a2, head.c2);
    }
    lo = fminf(lo, head.rlo);
    hi = fmaxf(hi, head.rhi);
    // further trips (B * N > 8 * kWG floats per array): two strides per trip, four loads in flight
    for (long k = tid0 + 2 * kWG; k < n4; k += 2 * kWG) {
      const long k2 = (k + kWG < n4) ? k + kWG : k;
      fold(x4[k], y4[k]);
      fold(x4[k2], y4[k2]);
    }
    for (long k = 4 * n4 + tid0; k < n_all; k += kWG) {
      lo = fminf(lo, fminf(x[k], y[k]));
      hi = fmaxf(hi, fmaxf(x[k], y[k]));
    }
    lo = eml::wave_min_dpp(lo);
    hi = eml::wave_max_dpp(hi);
    if ((tid0 & 63) == 0) {
      red_lo[tid0 >> 6] = lo;
      red_hi[tid0 >> 6] = hi;
    }
  }
  __syncthreads();
  // the f64 part (one log, one exp per schedule entry: a few hundred instructions) runs on wave 0 ONLY -- executed
  // redundantly by all 16 waves it held every SIMD for ~3 us of a 25 us kernel; the other waves go on to their staging
  if (tid0 < 64) {
    double d = diameter;
    if (diameter <= 0.0) {
      lo = red_lo[0];
      hi = red_hi[0];
#pragma unroll
      for (int w = 1; w < kWG / 64; ++w) {
        lo = fminf(lo, red_lo[w]);
        hi = fmaxf(hi, red_hi[w]);
      }
      d = (double)(hi - lo);  // f32 subtraction, then .item()
    }
    int cnt = 0;
    double start = 0.0;
    const double step = p_exp * log_scaling;
    if (d > 0.0) {
      start = p_exp * log(d);
      const double cntd = ceil((p_exp * log_blur - start) / step);  // numpy.arange length
      cnt = (cntd > 0.0) ? (int)fmin(cntd, (double)(EML_MAX_EPS - 2)) : 0;
    }
    if (tid0 < cnt + 2) {  // one schedule entry per lane (EML_MAX_EPS = 64 = one wave)
      double e;
      if (tid0 == 0) e = (p_exp == 2) ? d * d : pow(d, (double)p_exp);
      else if (tid0 == cnt + 1) e = (p_exp == 2) ? blur * blur : pow(blur, (double)p_exp);
      else e = exp(start + (tid0 - 1) * step);
      eps_l[tid0] = (float)e;
      if (blockIdx.x == 0 && eps_out) eps_out[tid0] = (float)e;
    }
    if (tid0 == 0) {
      *n_eps_l = cnt + 2;
      if (blockIdx.x == 0) {
        if (n_eps_out) *n_eps_out = cnt + 2;
        if (diameter_out) *diameter_out = (float)d;
      }
    }
  }
  if (kFinalBarrier) __syncthreads();   // otherwise the caller's next barrier publishes eps_l / n_eps_l
}

// threads per softmin group: cached kernel 512 (128 rows x 4 lanes), stream kernel 256 (row per thread)
template <bool kCached>
__global__ __launch_bounds__(kCached ? 1024 : 512) void sinkhorn_loop_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ M,
    const float* __restrict__ Mt, const float* __restrict__ alpha, const float* __restrict__ beta,
    double blur, double log_blur, double log_scaling, int p_exp, double diameter, const float* __restrict__ range_dev, float* __restrict__ eps_out,
    int* __restrict__ n_eps_out, float* __restrict__ diameter_out,
    float* __restrict__ work /* (8,B,N): duals a_x,b_y,a_y,b_x then E rows */, int B, int N) {
  constexpr int kGT = kCached ? 512 : 256;  // threads per softmin group
  constexpr int kWG = 2 * kGT;              // two groups per workgroup
  __shared__ float eps_l[EML_MAX_EPS];
  __shared__ int n_eps_l;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NP = round_up4(N) + kJPT;
  float* pts = smem;
  float* lw2 = pts + 2 * NP;
  float* h2 = lw2 + 2 * NP;
  float* potl = h2 + 4 * NP;
  float* Ml = potl + 2 * NP;

  const int b = blockIdx.x >> 1;
  const int role = blockIdx.x & 1;     // 0: (xx, yy)   1: (yx, xy)
  const int tid = threadIdx.x;
  const int gl = tid / kGT;            // local group 0/1; its columns are x for gl = 0 and y for gl = 1 in both roles
  const int g = 2 * role + gl;         // global problem id 0..3
  const int t = tid & (kGT - 1);
  const bool rows_x = (g == 0 || g == 3);
  const bool cols_x = (g == 0 || g == 2);
  const int consumer_l = role ? (1 - gl) : gl;  // local group whose h this potential feeds
  const float unif = 1.0f / (float)N;

  // ---- cached kernel: EVERY global read of the prologue is requested here, before anything waits -- this thread's
  // 16 chord-matrix elements, its point and weight, and (inside device_schedule) the diameter scan: one exposed
  // memory latency instead of four in a row (scan, points, weights, M), ~2.5 us of a 25 us kernel at cfg2
  if constexpr (kCached) EML_STAMP(0);   // kernel start
  constexpr int kMPT = kCached ? (4 * kCJ) * (4 * kCJ) / kWG : 1;   // 16 elements of M per thread
  float mreg[kMPT];
  float my_p = 0.f, my_w = unif;
  if constexpr (kCached) {
    const int nn = N * N;
    // (4-byte loads on purpose: staged as four float4 per thread the same 64 KB arrived 1.2 us later -- measured)
#pragma unroll
    for (int k = 0; k < kMPT; ++k) mreg[k] = M[min(tid + k * kWG, nn - 1)];
    if (tid < 2 * NP) {   // 2 * NP <= 384 < kWG
      const int which = tid >= NP, k = tid - which * NP;
      if (k < N) {
        my_p = (which == 0 ? x : y)[(size_t)b * N + k];
        const float* wp = which == 0 ? alpha : beta;
        if (wp) my_w = wp[(size_t)b * N + k];
      }
    }
  }
  const ScanHead scan_head = schedule_scan_begin<kWG>(x, y, B, N, diameter, range_dev);   // requested; folded in device_schedule
  if constexpr (kCached) {
    // staging runs WHILE the scan is in flight (the M / point loads were issued first, so they land first): points,
    // log-weights, h buffer 0 = log w of each group's columns (sweep 0 reads h = log w, potentials are zero:
    // sinkhorn_divergence.py:82-85), buffer 1 zeroed (the pads of both are read), and the chord matrix
    if (tid < 2 * NP) {
      const int k = tid - (tid >= NP) * NP;
      const float l = (k < N) ? ((my_w > 0.f) ? logf(my_w) : -100000.0f) : 0.f;  // sinkhorn_divergence.py:47-50
      pts[tid] = my_p;
      lw2[tid] = l * kLog2e;
      h2[tid] = l * kLog2e;        // h2[0][gl = which][k]
      h2[2 * NP + tid] = 0.f;
    }
    const int ldm = round_up4(N) + 4;
    {
      int r = tid / N, cc = tid - r * N;                 // element tid + k * kWG of M = (r, cc)
      const int dr = kWG / N, dc = kWG - dr * N;
      if (dc == 0) {   // N divides the workgroup size (N = 128, 64, ...): the column never changes, rows advance by dr
        float* dst = Ml + r * ldm + cc;
#pragma unroll
        for (int k = 0; k < kMPT; ++k)
          if (r + k * dr < N) dst[k * dr * ldm] = mreg[k];
      } else {
#pragma unroll
        for (int k = 0; k < kMPT; ++k) {
          if (r < N) Ml[r * ldm + cc] = mreg[k];
          r += dr;
          cc += dc;
          if (cc >= N) {
            cc -= N;
            ++r;
          }
        }
      }
    }
    EML_STAMP(1);   // this thread's staging written
  }
  device_schedule<kWG, !kCached>(scan_head, x, y, B, N, blur, log_blur, log_scaling, p_exp, diameter, eps_l, &n_eps_l, eps_out, n_eps_out,
                                 diameter_out);
  const float* eps_s = eps_l;

  // ---- stream kernel: stage points and log-weights; h buffer 0 = log w of each group's columns, buffer 1 zeroed
  if constexpr (!kCached) {
    for (int i = tid; i < 2 * NP; i += kWG) {
      const int which = i / NP, k = i - which * NP;
      float p = 0.f, l = 0.f;
      if (k < N) {
        p = (which == 0 ? x : y)[(size_t)b * N + k];
        const float* wp = which == 0 ? alpha : beta;
        const float w = wp ? wp[(size_t)b * N + k] : unif;
        l = (w > 0.f) ? logf(w) : -100000.0f;  // sinkhorn_divergence.py:47-50
      }
      pts[i] = p;
      lw2[i] = l * kLog2e;
      h2[i] = l * kLog2e;
      h2[2 * NP + i] = 0.f;
    }
  }

  const float* P = pts + (rows_x ? 0 : NP);
  const float* Q = pts + (cols_x ? 0 : NP);
  const float* lw2_rows = lw2 + (rows_x ? 0 : NP);
  const float* lw2_cols = lw2 + (cols_x ? 0 : NP);

  // ---- cached kernel: M -> LDS (coalesced), then this thread's costs -> registers
  // costs and exponents live in registers as float PAIRS so that the sweep's fma / subtract / add map onto
  // v_pk_fma_f32 / v_pk_add_f32 (two elements per VALU slot): the sweep is VALU-throughput-bound on the one CU that
  // holds a sample's coupled (yx, xy) problems
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f c2[kCJ / 2];
  int i = 0, jbase = 0;
  bool owner = false;
  if constexpr (kCached) {
    const int ldm = round_up4(N) + 4;
    __syncthreads();   // pts / lw2 / h2 / Ml / eps_l visible to every thread
    EML_STAMP(2);      // schedule computed, staging of every wave visible
    const int split = round_up4((N + 3) >> 2);   // columns per lane: 4 lanes share row i
    const int quarter = t & 3;
    i = t >> 2;
    jbase = quarter * split;
    const int cnt = (i < N) ? max(0, min(split, N - jbase)) : 0;
    owner = (quarter == 0) && (i < N);
    const int ic = min(i, N - 1);
    const float pi = P[ic];
    const float4* mrow = reinterpret_cast<const float4*>(Ml + ic * ldm + jbase);
    const float4* qrow = reinterpret_cast<const float4*>(Q + jbase);
#pragma unroll
    for (int q = 0; q < kCJ / 4; ++q) {
      const float4 mv = mrow[q];
      const float4 qv = qrow[q];
      // cost_ij on register pairs (v_pk_mul / v_pk_fma): 4 waves per SIMD build 32 costs per thread here
      const v2f p2 = v2f{pi, pi};
      const v2f qa = v2f{qv.x, qv.y}, qb = v2f{qv.z, qv.w};
      const v2f ca = ((p2 * p2 - 2.0f * (p2 * qa) + qa * qa) * 0.1f + v2f{mv.x, mv.y}) * 0.5f;
      const v2f cb = ((p2 * p2 - 2.0f * (p2 * qb) + qb * qb) * 0.1f + v2f{mv.z, mv.w}) * 0.5f;
      c2[2 * q + 0] = v2f{(4 * q + 0 < cnt) ? ca.x : kBig, (4 * q + 1 < cnt) ? ca.y : kBig};
      c2[2 * q + 1] = v2f{(4 * q + 2 < cnt) ? cb.x : kBig, (4 * q + 3 < cnt) ? cb.y : kBig};
    }
  }
  if constexpr (!kCached) __syncthreads();   // (the cached kernel's barrier sits after its M staging)
  const int n_eps = n_eps_l;
  if constexpr (kCached) EML_STAMP(3);   // costs in registers

  const size_t plane = (size_t)B * N;
  float* fin_out = work + (size_t)g * plane + (size_t)b * N;        // a_x | b_y | a_y | b_x
  float* e_out = work + (size_t)(4 + g) * plane + (size_t)b * N;    // E_i[q] of the last softmax

  float pot = 0.f;  // this row's potential (cached kernel)
  for (int s = 0; s < n_eps + 2; ++s) {
    const bool final_sweep = (s == n_eps + 1);
    const float eps = eps_s[(s == 0) ? 0 : min(s - 1, n_eps - 1)];
    const float eps_next = eps_s[min(s, n_eps - 1)];
    const float nie2 = -kLog2e / eps;
    const float k_next = kLog2e / eps_next;
    const float* hsrc = h2 + (s & 1) * 2 * NP + gl * NP;
    float* hdst = h2 + ((s + 1) & 1) * 2 * NP + consumer_l * NP;

    if constexpr (kCached) {
      const float4* hv4 = reinterpret_cast<const float4*>(hsrc + jbase);
      // four independent max / sum chains: with 2 waves per SIMD the sweep is latency-bound, not
      // throughput-bound, so instruction-level parallelism inside the wave is what shortens it
      // the exponents t_j = h_j - C_ij / eps are kept in registers between the max pass and the exp pass (the
      // sweep is VALU / transcendental-bound: recomputing them cost a second LDS read and an fma per element)
      v2f t2[kCJ / 2];
      const v2f n2 = v2f{nie2, nie2};
      v2f mA = v2f{-INFINITY, -INFINITY}, mB = mA;   // four independent max chains, as two pairs
#pragma unroll
      for (int q = 0; q < kCJ / 4; ++q) {
        const float4 hv = hv4[q];
        t2[2 * q + 0] = c2[2 * q + 0] * n2 + v2f{hv.x, hv.y};   // contracted: v_pk_fma_f32
        t2[2 * q + 1] = c2[2 * q + 1] * n2 + v2f{hv.z, hv.w};
        mA = __builtin_elementwise_max(mA, t2[2 * q + 0]);
        mB = __builtin_elementwise_max(mB, t2[2 * q + 1]);
      }
      float m = fmaxf(fmaxf(mA.x, mA.y), fmaxf(mB.x, mB.y));
      m = fmaxf(m, eml::lane_xor1(m));  // the 4 lanes of a row are a DPP quad
      m = fmaxf(m, eml::lane_xor2(m));
      const v2f mm = v2f{m, m};
      float sum = 0.f, tq = 0.f;
      if (!final_sweep) {
        v2f sA = v2f{0.f, 0.f}, sB = sA;   // s0, s1 | s2, s3 of the scalar form: same summation order
#pragma unroll
        for (int q = 0; q < kCJ / 4; ++q) {
          const v2f dA = t2[2 * q + 0] - mm, dB = t2[2 * q + 1] - mm;
          sA += v2f{__builtin_amdgcn_exp2f(dA.x), __builtin_amdgcn_exp2f(dA.y)};
          sB += v2f{__builtin_amdgcn_exp2f(dB.x), __builtin_amdgcn_exp2f(dB.y)};
        }
        sum = (sA.x + sA.y) + (sB.x + sB.y);
      } else {
        const float4* qrow = reinterpret_cast<const float4*>(Q + jbase);
#pragma unroll
        for (int q = 0; q < kCJ / 4; ++q) {
          const float4 qv = qrow[q];
          const v2f dA = t2[2 * q + 0] - mm, dB = t2[2 * q + 1] - mm;
          const float e0 = __builtin_amdgcn_exp2f(dA.x);
          const float e1 = __builtin_amdgcn_exp2f(dA.y);
          const float e2 = __builtin_amdgcn_exp2f(dB.x);
          const float e3 = __builtin_amdgcn_exp2f(dB.y);
          sum += (e0 + e1) + (e2 + e3);
          tq = fmaf(e0, qv.x, fmaf(e1, qv.y, fmaf(e2, qv.z, fmaf(e3, qv.w, tq))));
        }
        tq += eml::lane_xor1(tq);
        tq += eml::lane_xor2(tq);
      }
      sum += eml::lane_xor1(sum);
      sum += eml::lane_xor2(sum);
      // softmin = -eps * logsumexp (samples_loss.py:75-77), evaluated in base 2
      const float sm = -eps * kLn2 * (m + __builtin_amdgcn_logf(sum));
      if (final_sweep) {
        if (owner) {
          fin_out[i] = sm;
          e_out[i] = tq / sum;
        }
      } else {
        pot = (s == 0) ? sm : 0.5f * (pot + sm);  // symmetrised update, sinkhorn_divergence.py:96-97
        if (owner) hdst[i] = fmaf(pot, k_next, lw2_rows[i]);
      }
    } else {
      for (int r = t; r < N; r += kGT) {
        const float pi = P[r];
        float m = -INFINITY;
#pragma unroll 4
        for (int j = 0; j < N; ++j)
          m = fmaxf(m, fmaf(cost_ij(pi, Q[j], Mt[(size_t)j * N + r]), nie2, hsrc[j]));
        float sum = 0.f, tq = 0.f;
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
          const float qj = Q[j];
          const float e =
              __builtin_amdgcn_exp2f(fmaf(cost_ij(pi, qj, Mt[(size_t)j * N + r]), nie2, hsrc[j] - m));
          sum += e;
          tq = fmaf(e, qj, tq);
        }
        const float sm = -eps * kLn2 * (m + __builtin_amdgcn_logf(sum));
        if (final_sweep) {
          fin_out[r] = sm;
          e_out[r] = tq / sum;
        } else {
          const float pr = (s == 0) ? sm : 0.5f * (potl[gl * NP + r] + sm);
          potl[gl * NP + r] = pr;
          hdst[r] = fmaf(pr, k_next, lw2_rows[r]);
        }
      }
    }
    __syncthreads();
    if constexpr (kCached) EML_STAMP(4 + s);   // sweep s done
  }
}

// ---- "tiled" kernel, 128 < N <= 512 (N % 4 == 0): the N x N cost blocks do not fit registers any more, so the chord
// matrix streams through LDS in column tiles (256 rows x 32 columns, or 512 x 16: 8192 floats, double-buffered) that
// BOTH softmin problems of the workgroup consume; a row is shared by LPR lanes, each folding its 16 columns of a tile
// into a running (max, sum) pair -- an online log-sum-exp, one pass per sweep instead of the stream kernel's two passes
// over an L2-resident Mt.  The tile sequence is the same in every sweep, so the next tile's global loads are always in
// flight during the current tile's arithmetic, across sweep boundaries too.  One LDS-only barrier per tile.
template <int LPR /* lanes per row: 2 (N <= 256) or 1 (N <= 512) */>
__global__ __launch_bounds__(1024) void sinkhorn_loop_tiled_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ M,
    const float* __restrict__ alpha, const float* __restrict__ beta, double blur, double log_blur, double log_scaling,
    int p_exp,
    double diameter, const float* __restrict__ range_dev, float* __restrict__ eps_out, int* __restrict__ n_eps_out, float* __restrict__ diameter_out,
    float* __restrict__ work, int B, int N, const int* __restrict__ only_if) {
  // `only_if` (device, may be NULL): the RESCUE launch behind the split kernel -- it returns at once unless the split
  // kernel raised its status word (a slice never saw its partners), in which case it recomputes the whole batch here
  if (only_if && __hip_atomic_load(only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  constexpr int kGT = 512, kWG = 1024;
  constexpr int NR = 512 / LPR;            // row capacity of a softmin group
  constexpr int TJ = 16 * LPR;             // tile columns; every lane folds 16 of them
  constexpr int TS = TJ + 4;               // LDS row stride of a tile
  __shared__ float eps_l[EML_MAX_EPS];
  __shared__ int n_eps_l;
  device_schedule<kWG>(schedule_scan_begin<kWG>(x, y, B, N, diameter, range_dev), x, y, B, N, blur, log_blur, log_scaling, p_exp, diameter,
                       eps_l, &n_eps_l, eps_out, n_eps_out, diameter_out);
  const float* eps_s = eps_l;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NP = round_up4(N) + kJPT;
  float* pts = smem;
  float* lw2 = pts + 2 * NP;
  float* h2 = lw2 + 2 * NP;
  float* qq = h2 + 4 * NP;                 // [2][NP]: 0.05 * point^2 (the column-only term of the cost), pads zero
  float* Mtile = qq + 2 * NP;              // [2][NR][TS]

  const int b = blockIdx.x >> 1, role = blockIdx.x & 1, tid = threadIdx.x;
  const int gl = tid / kGT, g = 2 * role + gl, t = tid & (kGT - 1);
  const bool rows_x = (g == 0 || g == 3), cols_x = (g == 0 || g == 2);
  const int consumer_l = role ? (1 - gl) : gl;
  const int n_eps = n_eps_l;

  const float unif = 1.0f / (float)N;
  for (int i = tid; i < 2 * NP; i += kWG) {
    const int which = i / NP, k = i - which * NP;
    float p = 0.f, l = 0.f;
    if (k < N) {
      p = (which == 0 ? x : y)[(size_t)b * N + k];
      const float* wp = which == 0 ? alpha : beta;
      const float w = wp ? wp[(size_t)b * N + k] : unif;
      l = (w > 0.f) ? logf(w) : -100000.0f;
    }
    pts[i] = p;
    qq[i] = 0.05f * p * p;
    lw2[i] = l * kLog2e;
  }
  for (int i = tid; i < 4 * NP; i += kWG) h2[i] = 0.f;
  __syncthreads();
  const float* P = pts + (rows_x ? 0 : NP);
  const float* Q = pts + (cols_x ? 0 : NP);
  const float* lw2_rows = lw2 + (rows_x ? 0 : NP);
  const float* lw2_cols = lw2 + (cols_x ? 0 : NP);
  const float* QQ = qq + (cols_x ? 0 : NP);
  // the dual vector travels as A_j = h_j + n * .05 q_j^2 (n = -log2(e)/eps of the sweep that READS it): the column-only
  // term of the exponent is folded in by whoever writes h, once per column and sweep, not once per element
  {
    const float k0 = kLog2e / eps_s[0];
    for (int k = t; k < N; k += kGT) h2[gl * NP + k] = fmaf(-k0, QQ[k], lw2_cols[k]);
  }

  const int i = t / LPR, part = t % LPR;
  const bool owner = part == 0 && i < N;
  const int ic = min(i, N - 1);
  const float pi = P[ic];
  const int ntiles = (N + TJ - 1) / TJ;
  // staging role: row srow, 8 consecutive columns of the tile
  const int srow = tid / (TJ / 8), sq = tid % (TJ / 8);
  const float* mrow = M + (size_t)min(srow, N - 1) * N;
  float4 s0, s1;
  auto stage_load = [&](int tile) {   // unconditional, clamped: N % 4 == 0, so a float4 is entirely inside or outside
    const int j = tile * TJ + 8 * sq;
    s0 = *reinterpret_cast<const float4*>(mrow + min(j, N - 4));
    s1 = *reinterpret_cast<const float4*>(mrow + min(j + 4, N - 4));
  };
  auto stage_commit = [&](int buf) {
    float* d = Mtile + (size_t)buf * NR * TS + srow * TS + 8 * sq;
    *reinterpret_cast<float4*>(d) = s0;
    *reinterpret_cast<float4*>(d + 4) = s1;
  };
  stage_load(0);
  stage_commit(0);
  __syncthreads();

  const size_t plane = (size_t)B * N;
  float* fin_out = work + (size_t)g * plane + (size_t)b * N;
  float* e_out = work + (size_t)(4 + g) * plane + (size_t)b * N;
  float pot = 0.f;
  int buf = 0;
  for (int s = 0; s < n_eps + 2; ++s) {
    const bool final_sweep = (s == n_eps + 1);
    const float eps = eps_s[(s == 0) ? 0 : min(s - 1, n_eps - 1)];
    const float eps_next = eps_s[min(s, n_eps - 1)];
    const float nie2 = -kLog2e / eps, k_next = kLog2e / eps_next;
    const float* hsrc = h2 + (s & 1) * 2 * NP + gl * NP;
    float* hdst = h2 + ((s + 1) & 1) * 2 * NP + consumer_l * NP;
    float m_run = -INFINITY, s_run = 0.f, tq_run = 0.f;
    for (int tile = 0; tile < ntiles; ++tile) {
      stage_load(tile + 1 < ntiles ? tile + 1 : 0);   // wraps: the first tile of the next sweep
      __builtin_amdgcn_sched_barrier(0);
      const int j0 = tile * TJ + 16 * part;
      const float* mt = Mtile + (size_t)buf * NR * TS + min(i, NR - 1) * TS + 16 * part;
      // the exponents t_j = h_j - C_ij / eps with C_ij = .05 p^2 - .1 p q + .05 q^2 + .5 m expanded around its column- and
      // row-only parts:  t_j = [h_j + n*.05 q_j^2] + (-.1 n p_i) q_j + (.5 n) m_ij + n*.05 p_i^2,  n = -log2(e)/eps.  The last
      // term is constant along the row: it is added to the row's maximum after the sweep (row_shift), so an element costs
      // TWO fused multiply-adds on register pairs (v_pk_fma_f32) -- the bracket arrives ready-made in the dual vector --
      // instead of six operations for cost + exponent, and three LDS reads (M tile, q, A) instead of four
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f tv[8];
      const v2f dd = v2f{-0.1f * nie2 * pi, -0.1f * nie2 * pi}, ee = v2f{0.5f * nie2, 0.5f * nie2};
      v2f mA = v2f{-INFINITY, -INFINITY}, mB = mA;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 mv = *reinterpret_cast<const float4*>(mt + 4 * u);
        const float4 qv = *reinterpret_cast<const float4*>(Q + j0 + 4 * u);      // pads of pts / qq / h2 are zero-filled
        const float4 hv = *reinterpret_cast<const float4*>(hsrc + j0 + 4 * u);   // A_j (see the h2 initialisation)
        const bool jv = j0 + 4 * u < N;
        const v2f b0 = dd * v2f{qv.x, qv.y} + v2f{hv.x, hv.y}, b1 = dd * v2f{qv.z, qv.w} + v2f{hv.z, hv.w};
        const v2f ninf = v2f{-INFINITY, -INFINITY};
        tv[2 * u + 0] = jv ? ee * v2f{mv.x, mv.y} + b0 : ninf;
        tv[2 * u + 1] = jv ? ee * v2f{mv.z, mv.w} + b1 : ninf;
        mA = __builtin_elementwise_max(mA, tv[2 * u + 0]);
        mB = __builtin_elementwise_max(mB, tv[2 * u + 1]);
      }
      const float mx = fmaxf(fmaxf(mA.x, mA.y), fmaxf(mB.x, mB.y));
      const float m_new = fmaxf(m_run, mx);   // finite from the first tile on (its first columns are always valid)
      const float rescale = __builtin_amdgcn_exp2f(m_run - m_new);
      const v2f mm = v2f{m_new, m_new};
      float sA = 0.f, sB = 0.f, qA = 0.f, qB = 0.f;
      if (!final_sweep) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const v2f d = tv[u] - mm;
          sA += __builtin_amdgcn_exp2f(d.x);
          sB += __builtin_amdgcn_exp2f(d.y);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const v2f d = tv[u] - mm;
          const float e0 = __builtin_amdgcn_exp2f(d.x), e1 = __builtin_amdgcn_exp2f(d.y);
          sA += e0;
          sB += e1;
          qA = fmaf(e0, Q[j0 + 2 * u], qA);
          qB = fmaf(e1, Q[j0 + 2 * u + 1], qB);
        }
      }
      s_run = fmaf(s_run, rescale, sA + sB);
      tq_run = fmaf(tq_run, rescale, qA + qB);
      m_run = m_new;
      __builtin_amdgcn_sched_barrier(0);
      stage_commit(buf ^ 1);
      eml::lds_barrier();
      buf ^= 1;
    }
    float m = m_run, sum = s_run, tq = tq_run;
    if constexpr (LPR == 2) {   // the two lanes of a row are DPP neighbours
      const float mo = eml::lane_xor1(m_run), so = eml::lane_xor1(s_run), qo = eml::lane_xor1(tq_run);
      m = fmaxf(m_run, mo);
      const float fa = __builtin_amdgcn_exp2f(m_run - m), fb = __builtin_amdgcn_exp2f(mo - m);
      sum = s_run * fa + so * fb;
      tq = tq_run * fa + qo * fb;
    }
    const float row_shift = 0.05f * nie2 * pi * pi;   // the row-only term of the exponents, left out of the tiles above
    const float sm = -eps * kLn2 * ((m + row_shift) + __builtin_amdgcn_logf(sum));
    if (final_sweep) {
      if (owner) {
        fin_out[i] = sm;
        e_out[i] = tq / sum;
      }
    } else {
      pot = (s == 0) ? sm : 0.5f * (pot + sm);
      // this row is a COLUMN of the group that reads hdst: fold its .05 p^2 term in with that sweep's n = -k_next
      if (owner) hdst[i] = fmaf(pot, k_next, lw2_rows[i]) - k_next * (0.05f * pi * pi);
    }
    __syncthreads();
  }
}

// ---- "split" kernel: small batches at N >= 192 (BASELINE configs[4]: 16 samples x 256 anchors per GPU).  With one
// workgroup pair per sample only 2 * B of the 256 CUs work and a sweep is bound by the exponentials of ONE CU (tiled kernel:
// 11.3 us per sweep at N = 256).  Here the ROWS of a sample's problem pair are split over S workgroups (S = 8 or 4: 2 * B * S
// workgroups, one per CU at B = 16): workgroup (b, role, slice) owns R = N / S rows of both problems of its role and ALL N
// columns.  Its R x N slice of the chord matrix is loaded ONCE into LDS (33 KB at N = 256; no tile streaming), eight lanes
// share a row and fold N / 8 columns each from registers, and after every sweep the slices exchange the dual vector (N
// floats per problem) through global memory as 8-byte {epoch, value} granules (cdna_hip_programming.md G16, form R2: the
// data is the flag -- relaxed agent-scope 8-byte stores and polls, no fences; double-buffered by sweep parity, so a
// workgroup that runs ahead never overwrites a granule a slower one still waits for; the launcher zeroes the exchange
// buffer with a memset node in front of the kernel).  The exchange is confined to the S slices of one (sample, role) --
// S consecutive workgroup ids -- and needs THOSE co-resident (measured: forced onto a stream masked down to 32 CUs the
// 256-workgroup launch still completes, the dispatcher places consecutive ids together; on 2 CUs, 6 resident workgroups,
// it cannot).  The launcher only takes this path when every workgroup gets a CU of its own on the CUs the STREAM may use
// (that is what makes the split pay, and far more than residency needs), but residency cannot be guaranteed from the
// host (another stream's kernel, a second process on the device, ...), so it is also enforced on the device: polls are
// bounded by wall-clock time (50 ms -- a healthy exchange takes ~1 us); a workgroup that never sees its partners raises
// the call's STATUS word (device int, zeroed by the launcher's memset), every other workgroup notices the word in its own
// poll loop or at its start and leaves, and the launcher has ALREADY enqueued the tiled kernel behind this one, gated on
// that word: it recomputes the whole batch (an empty ~2 us launch otherwise).  The caller never sees a NaN; the status
// word tells it that the slow path ran (eml_sinkhorn_fwd_ex_f32).
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int kSplitLPR = 8;        // lanes per row
constexpr long long kSpinTicks = 5000000LL;   // 50 ms of the 100 MHz wall clock

template <int CPL /* columns per lane = N / 8 */, int R /* rows per workgroup = N / S */>
__global__ __launch_bounds__(16 * R) void sinkhorn_loop_split_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ M,
    const float* __restrict__ alpha, const float* __restrict__ beta, double blur, double log_blur, double log_scaling,
    int p_exp, double diameter, const float* __restrict__ range_dev, float* __restrict__ eps_out,
    int* __restrict__ n_eps_out, float* __restrict__ diameter_out, float* __restrict__ work,
    unsigned long long* __restrict__ exch, int B, int S, int* __restrict__ status, int test_stall) {
  constexpr int N = CPL * kSplitLPR, kWG = 16 * R, kGT = 8 * R, LDM = N + 32;   // LDM = 32 mod 64: two rows cover all banks
  static_assert(CPL % 4 == 0 && N % 64 == 0 && kWG <= 1024, "split kernel geometry");
  __shared__ float eps_l[EML_MAX_EPS];
  __shared__ int n_eps_l;
  __shared__ int failed_l;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* pts = smem;               // [2][N] x, y
  float* lw2 = pts + 2 * N;        // [2][N] log2(e) * log-weights
  float* qq = lw2 + 2 * N;         // [2][N] .05 * point^2
  float* h2 = qq + 2 * N;          // [2 parity][2 problems][N]
  float* Ml = h2 + 4 * N;          // [R][LDM]
  const int tid = threadIdx.x;
  const int slice = blockIdx.x % S, br = blockIdx.x / S, b = br >> 1, role = br & 1;
  const int r0 = slice * R;
  const ScanHead scan_head = schedule_scan_begin<kWG>(x, y, B, N, diameter, range_dev);
  // this workgroup's rows of the chord matrix: requested before anything waits, committed below
  constexpr int kM4 = R * N / 4 / kWG;     // float4 per thread: R * N / (64 R) = N / 64
  float4 mreg[kM4];
#pragma unroll
  for (int k = 0; k < kM4; ++k) {
    const int e = tid + k * kWG, row = e / (N / 4), c4 = e % (N / 4);
    mreg[k] = *reinterpret_cast<const float4*>(M + (size_t)(r0 + row) * N + 4 * c4);
  }
  // a workgroup that starts after a partner already gave up leaves at once (after the schedule's barrier)
  if (tid == 0) failed_l = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float unif = 1.0f / (float)N;
  for (int i = tid; i < 2 * N; i += kWG) {
    const int which = i / N, k = i - which * N;
    const float p = (which == 0 ? x : y)[(size_t)b * N + k];
    const float* wp = which == 0 ? alpha : beta;
    const float w = wp ? wp[(size_t)b * N + k] : unif;
    pts[i] = p;
    qq[i] = 0.05f * p * p;
    lw2[i] = ((w > 0.f) ? logf(w) : -100000.0f) * kLog2e;   // sinkhorn_divergence.py:47-50
  }
#pragma unroll
  for (int k = 0; k < kM4; ++k) {
    const int e = tid + k * kWG, row = e / (N / 4), c4 = e % (N / 4);
    *reinterpret_cast<float4*>(Ml + row * LDM + 4 * c4) = mreg[k];
  }
  device_schedule<kWG>(scan_head, x, y, B, N, blur, log_blur, log_scaling, p_exp, diameter, eps_l, &n_eps_l,
                       (slice == 0) ? eps_out : nullptr, (slice == 0) ? n_eps_out : nullptr,
                       (slice == 0) ? diameter_out : nullptr);   // ends with a barrier: LDS staging above is visible
  const float* eps_s = eps_l;
  const int n_eps = n_eps_l;
  if (failed_l) return;   // workgroup-uniform (read after a barrier): the rescue launch recomputes the batch

  const int gl = tid / kGT, g = 2 * role + gl, t = tid - gl * kGT;
  const bool rows_x = (g == 0 || g == 3), cols_x = (g == 0 || g == 2);
  const int consumer_l = role ? (1 - gl) : gl;
  const float* P = pts + (rows_x ? 0 : N);
  const float* Q = pts + (cols_x ? 0 : N);
  const float* lw2_rows = lw2 + (rows_x ? 0 : N);
  const float* lw2_cols = lw2 + (cols_x ? 0 : N);
  const float* QQ = qq + (cols_x ? 0 : N);
  {   // sweep 0 reads A_j = log w_j - k0 * .05 q_j^2 (potentials are zero: sinkhorn_divergence.py:82-85)
    const float k0 = kLog2e / eps_s[0];
    for (int k = t; k < N; k += kGT) h2[gl * N + k] = fmaf(-k0, QQ[k], lw2_cols[k]);
  }
  __syncthreads();

  const int il = t / kSplitLPR, part = t % kSplitLPR, i = r0 + il;
  const bool owner = part == 0;
  const float pi = P[i];
  const float* mrow = Ml + il * LDM;
  const size_t plane = (size_t)B * N;
  float* fin_out = work + (size_t)g * plane + (size_t)b * N;
  float* e_out = work + (size_t)(4 + g) * plane + (size_t)b * N;
  gu64* xg = (gu64*)exch;   // [2 parity][B][2 roles][2 problems][N]
  const size_t xpar = (size_t)B * 4 * N, xbase = (size_t)br * 2 * N;
  float pot = 0.f;
  for (int s = 0; s < n_eps + 2; ++s) {
    const bool final_sweep = (s == n_eps + 1);
    const float eps = eps_s[(s == 0) ? 0 : min(s - 1, n_eps - 1)];
    const float eps_next = eps_s[min(s, n_eps - 1)];
    const float nie2 = -kLog2e / eps, k_next = kLog2e / eps_next;
    const float* hsrc = h2 + (s & 1) * 2 * N + gl * N;
    // exponents t_j = [A_j + (-.1 n p_i) q_j] + (.5 n) m_ij (+ the row-only term, added to the maximum afterwards), as in the
    // tiled kernel; lane `part` owns the float4 column chunks part, part + 8, ...: the 8 lanes of a row read 32 consecutive
    // floats per step and two rows (LDM = 32 mod 64) cover all 64 banks
    typedef float v2f __attribute__((ext_vector_type(2)));
    // CPL <= 48: the exponents stay in registers between the max pass and the exp pass; wider rows (N = 448, 512 at
    // 896 / 1024 threads = 128 VGPRs per lane) recompute them from LDS instead of spilling 64 of them to scratch
    constexpr bool kKeep = CPL <= 48;
    constexpr int kUnroll = kKeep ? CPL / 4 : 4;
    v2f tv[kKeep ? CPL / 2 : 2];
    const v2f dd = v2f{-0.1f * nie2 * pi, -0.1f * nie2 * pi}, ee = v2f{0.5f * nie2, 0.5f * nie2};
    auto expo = [&](int u, v2f& a, v2f& c) {
      const int j = 4 * (part + kSplitLPR * u);
      const float4 mv = *reinterpret_cast<const float4*>(mrow + j);
      const float4 qv = *reinterpret_cast<const float4*>(Q + j);
      const float4 hv = *reinterpret_cast<const float4*>(hsrc + j);
      a = ee * v2f{mv.x, mv.y} + (dd * v2f{qv.x, qv.y} + v2f{hv.x, hv.y});
      c = ee * v2f{mv.z, mv.w} + (dd * v2f{qv.z, qv.w} + v2f{hv.z, hv.w});
    };
    v2f mA = v2f{-INFINITY, -INFINITY}, mB = mA;
#pragma unroll kUnroll
    for (int u = 0; u < CPL / 4; ++u) {
      v2f a, c;
      expo(u, a, c);
      if constexpr (kKeep) {
        tv[2 * u + 0] = a;
        tv[2 * u + 1] = c;
      }
      mA = __builtin_elementwise_max(mA, a);
      mB = __builtin_elementwise_max(mB, c);
    }
    float m = fmaxf(fmaxf(mA.x, mA.y), fmaxf(mB.x, mB.y));
    m = fmaxf(m, eml::lane_xor1(m));
    m = fmaxf(m, eml::lane_xor2(m));
    m = fmaxf(m, eml::dpp_mov<0x141>(m));   // row_half_mirror: the other quad of this row's 8 lanes
    const v2f mm = v2f{m, m};
    float sA = 0.f, sB = 0.f, qA = 0.f, qB = 0.f;
#pragma unroll kUnroll
    for (int u = 0; u < CPL / 4; ++u) {
      v2f a, c;
      if constexpr (kKeep) {
        a = tv[2 * u + 0];
        c = tv[2 * u + 1];
      } else {
        expo(u, a, c);
      }
      const v2f d0 = a - mm, d1 = c - mm;
      const float e0 = __builtin_amdgcn_exp2f(d0.x), e1 = __builtin_amdgcn_exp2f(d0.y);
      const float e2 = __builtin_amdgcn_exp2f(d1.x), e3 = __builtin_amdgcn_exp2f(d1.y);
      sA += e0 + e2;
      sB += e1 + e3;
      if (final_sweep) {   // expectation of the column points under the row's softmax: the analytic gradient
        const float4 qv = *reinterpret_cast<const float4*>(Q + 4 * (part + kSplitLPR * u));
        qA = fmaf(e0, qv.x, fmaf(e2, qv.z, qA));
        qB = fmaf(e1, qv.y, fmaf(e3, qv.w, qB));
      }
    }
    float sum = sA + sB, tq = qA + qB;
    sum += eml::lane_xor1(sum);
    sum += eml::lane_xor2(sum);
    sum += eml::dpp_mov<0x141>(sum);
    const float row_shift = 0.05f * nie2 * pi * pi;
    const float sm = -eps * kLn2 * ((m + row_shift) + __builtin_amdgcn_logf(sum));
    if (final_sweep) {
      tq += eml::lane_xor1(tq);
      tq += eml::lane_xor2(tq);
      tq += eml::dpp_mov<0x141>(tq);
      if (owner) {
        fin_out[i] = sm;
        e_out[i] = tq / sum;
      }
      break;
    }
    pot = (s == 0) ? sm : 0.5f * (pot + sm);
    // ---- exchange: this row is COLUMN i of the problem that reads it next sweep (its .05 p^2 term folded in with that
    // sweep's n = -k_next); every workgroup of the (sample, role) then collects all 2 * N granules of the new parity
    const unsigned epoch = (unsigned)(s + 1);
    gu64* xw = xg + ((s + 1) & 1) * xpar + xbase;
    // test_stall (EML_SINKHORN_TEST_STALL): slice 3 of every group never publishes -- what its partners would see if it were
    // not resident -- so that the give-up and the rescue can be exercised on an idle device
    if (owner && !(test_stall && slice == 3)) {
      const float hv = fmaf(pot, k_next, lw2_rows[i]) - k_next * (0.05f * pi * pi);
      __hip_atomic_store(xw + consumer_l * N + i, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, hv),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float* hdst = h2 + ((s + 1) & 1) * 2 * N;
    for (int k = tid; k < 2 * N; k += kWG) {
      long long t0 = 0;
      unsigned spins = 0;
      for (;;) {
        const unsigned long long v = __hip_atomic_load(xw + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == epoch) {
          hdst[k] = __builtin_bit_cast(float, (unsigned)v);
          break;
        }
        if ((++spins & 1023u) == 0) {   // bounded by wall-clock time, checked rarely; a partner's give-up ends the wait too
          const long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          const bool timed_out = now - t0 > kSpinTicks;
          if (timed_out || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            if (timed_out) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            failed_l = 1;
            hdst[k] = 0.f;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (failed_l) {   // workgroup-uniform after the barrier.  Poison this slice's rows: if the rescue launch behind this
      if (owner) {    // kernel did not run for any reason, the caller gets NaN, never a plausible wrong number
        fin_out[i] = NAN;
        e_out[i] = NAN;
      }
      return;
    }
  }
}

// loss_b = <alpha, b_x - a_x> + <beta, a_y - b_y>  (sinkhorn_divergence.py:65-69) and the
// analytic backward of the last extrapolation: the gradient reaches x only through the final
// xx / xy softmins and only through the cost's first argument (utils.py:88), and since the
// softmax rows sum to one,  dL_b/dx_i = alpha_i * 0.1 * (E^xx_i[x] - E^xy_i[y])  (same for y).
__global__ __launch_bounds__(256) void sinkhorn_finish_kernel(
    const float* __restrict__ work, const float* __restrict__ alpha, const float* __restrict__ beta,
    float* __restrict__ loss, float* __restrict__ gx, float* __restrict__ gy, int B, int N) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t plane = (size_t)B * N, row = (size_t)b * N;
  const float unif = 1.0f / (float)N;
  float part = 0.f;
  for (int k = tid; k < N; k += 256) {
    const size_t o = row + k;
    const float a_x = work[o], b_y = work[plane + o], a_y = work[2 * plane + o], b_x = work[3 * plane + o];
    const float al = alpha ? alpha[o] : unif, be = beta ? beta[o] : unif;
    part += al * (b_x - a_x) + be * (a_y - b_y);
    if (gx) gx[o] = al * 0.1f * (work[4 * plane + o] - work[7 * plane + o]);
    if (gy) gy[o] = be * 0.1f * (work[5 * plane + o] - work[6 * plane + o]);
  }
  part = eml::wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) loss[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Chord matrix M_ij = ||a_i - a_j||_2 on f32 anchors (geomloss/utils.py:67-76).
__global__ __launch_bounds__(256) void anchor_cost_kernel(const float* __restrict__ a,
                                                          float* __restrict__ M, int N) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * N) return;
  const int i = idx / N, j = idx - i * N;
  const float dx = a[3 * i] - a[3 * j], dy = a[3 * i + 1] - a[3 * j + 1], dz = a[3 * i + 2] - a[3 * j + 2];
  M[idx] = sqrtf(dx * dx + dy * dy + dz * dz);
}

// Diameter (range of x U y) + epsilon schedule, all on the device.
__global__ __launch_bounds__(1024) void schedule_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ y, long n,
                                                        double blur, double scaling, int p,
                                                        double diameter, const float* __restrict__ range_dev, float* __restrict__ eps_out,
                                                        int* __restrict__ n_eps_out,
                                                        float* __restrict__ diameter_out) {
  __shared__ float red_min[16], red_max[16];
  const int tid = threadIdx.x;
  float lo = INFINITY, hi = -INFINITY;
  if (diameter <= 0.0) {
    for (long k = tid; k < n; k += 1024) {
      const float a = x[k], c = y[k];
      lo = fminf(lo, fminf(a, c));
      hi = fmaxf(hi, fmaxf(a, c));
    }
    lo = eml::wave_min(lo);
    hi = eml::wave_max(hi);
    if ((tid & 63) == 0) {
      red_min[tid >> 6] = lo;
      red_max[tid >> 6] = hi;
    }
  }
  __syncthreads();
  if (tid != 0) return;
  double d = diameter;
  if (diameter <= 0.0) {
    for (int w = 0; w < 16; ++w) {
      lo = fminf(lo, red_min[w]);
      hi = fmaxf(hi, red_max[w]);
    }
    if (range_dev) {   // the other ranks' range (all-reduced by the caller)
      lo = fminf(lo, range_dev[0]);
      hi = fmaxf(hi, range_dev[1]);
    }
    d = (double)(hi - lo);  // f32 subtraction, then .item(): sinkhorn_divergence.py:15
  }
  *diameter_out = (float)d;
  // sinkhorn_divergence.py:21-25, in f64 like numpy
  int k = 0;
  eps_out[k++] = (float)((p == 2) ? d * d : pow(d, (double)p));
  if (d > 0.0) {
    const double start = p * log(d), stop = p * log(blur), step = p * log(scaling);
    const double cntd = ceil((stop - start) / step);  // numpy.arange length
    const int cnt = (cntd > 0.0) ? (int)fmin(cntd, (double)(EML_MAX_EPS - 2)) : 0;
    for (int e = 0; e < cnt; ++e) eps_out[k++] = (float)exp(start + e * step);
  }
  eps_out[k++] = (float)((p == 2) ? blur * blur : pow(blur, (double)p));
  *n_eps_out = k;
  for (; k < EML_MAX_EPS; ++k) eps_out[k] = 0.f;
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ gloss,
                                                         const float* __restrict__ gunit,
                                                         float* __restrict__ gout, int B, int N) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * N) return;
  gout[idx] = gloss[idx / N] * gunit[idx];
}

}  // namespace

extern "C" int eml_emd_anchor_cost_f32(const float* anchors, float* M, int N, eml_stream_t stream) {
  if (!anchors || !M || N < 1 || N > 16384) return eml::fail(EML_EINVAL, "eml_emd_anchor_cost_f32: bad arguments");
  hipLaunchKernelGGL(anchor_cost_kernel, dim3((N * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, anchors, M, N);
  return eml::check_launch("eml_emd_anchor_cost_f32");
}

extern "C" int eml_sinkhorn_schedule_f32(const float* x, const float* y, long n, double blur,
                                         double scaling, int p, double diameter, const float* range_lo_hi,
                                         float* eps_out, int* n_eps_out, float* diameter_out, eml_stream_t stream) {
  if (!eps_out || !n_eps_out || !diameter_out) return eml::fail(EML_EINVAL, "eml_sinkhorn_schedule_f32: null output");
  if (diameter <= 0.0 && (!x || !y || n < 1))
    return eml::fail(EML_EINVAL, "eml_sinkhorn_schedule_f32: need x, y, n>=1 when diameter is not given");
  if (!(blur > 0.0) || !(scaling > 0.0 && scaling < 1.0) || p < 1)
    return eml::fail(EML_EINVAL, "eml_sinkhorn_schedule_f32: need blur>0, 0<scaling<1, p>=1");
  hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, n, blur, scaling, p,
                     diameter, range_lo_hi, eps_out, n_eps_out, diameter_out);
  return eml::check_launch("eml_sinkhorn_schedule_f32");
}

// (8,B,N) floats of duals / expectation rows + the split kernel's exchange buffer: [2 parity][B][2 roles][2 problems][N]
// 8-byte granules = 16*B*N floats + 4 status words (word 0: the split kernel gave up and the tiled kernel recomputed)
extern "C" size_t eml_sinkhorn_work_floats(int B, int N) { return (size_t)24 * B * N + 4; }

namespace {
int device_cu_count() {
  static std::atomic<int> cached[eml::kMaxDevices];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& c = cached[dev & (eml::kMaxDevices - 1)];
  int n = c.load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 1;
    c.store(n, std::memory_order_relaxed);
  }
  return n;
}

// CUs a kernel launched on `stream` may occupy: the stream's own CU mask (hipExtStreamCreateWithCUMask) or the process-wide
// one (ROC_GLOBAL_CU_MASK) -- hipExtStreamGetCUMask reports whichever applies -- never more than the device has.
int stream_cu_count(hipStream_t stream) {
  const int dev_cus = device_cu_count();
  uint32_t mask[16] = {0};   // 512 CUs
  if (hipExtStreamGetCUMask(stream, 16, mask) != hipSuccess) {
    (void)hipGetLastError();
    return dev_cus;
  }
  int n = 0;
  for (int w = 0; w < 16; ++w) n += __builtin_popcount(mask[w]);
  return (n < 1 || n > dev_cus) ? dev_cus : n;
}

// resident workgroups of the split kernel per CU (registers, LDS, wave slots), cached per (instance, device)
template <typename K>
int split_occupancy(K kernel, int block, size_t lds, std::atomic<int>* cache) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& c = cache[dev & (eml::kMaxDevices - 1)];
  int n = c.load(std::memory_order_relaxed);
  if (n == 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block, lds) != hipSuccess) {
      (void)hipGetLastError();
      occ = 0;
    }
    n = occ > 0 ? occ : -1;   // -1: cannot be made resident at all
    c.store(n, std::memory_order_relaxed);
  }
  return n > 0 ? n : 0;
}
}  // namespace

namespace {
size_t split_lds_bytes(int N, int R) { return (size_t)(10 * N + R * (N + 32)) * sizeof(float); }

// LDS-tiled kernel: chord-matrix column tiles (8192 floats, double-buffered) shared by both problems of a workgroup.
// `only_if`: see the kernel (the rescue launch behind the split kernel).
void launch_tiled(const float* x, const float* y, const float* M, const float* alpha, const float* beta, double blur,
                  double log_blur, double log_scaling, int p, double diameter, const float* range_lo_hi, float* eps_out,
                  int* n_eps_out, float* diameter_out, float* work, int B, int N, const int* only_if, hipStream_t stream) {
  const int NP = round_up4(N) + kJPT;
  const int lpr = N <= 256 ? 2 : 1;
  const size_t lds = (size_t)(10 * NP + 2 * (512 / lpr) * (16 * lpr + 4)) * sizeof(float);
  if (lpr == 2) {
    EML_ENSURE_LDS((&sinkhorn_loop_tiled_kernel<2>), lds);
    hipLaunchKernelGGL(sinkhorn_loop_tiled_kernel<2>, dim3(2 * B), dim3(1024), lds, stream, x, y, M, alpha, beta, blur,
                       log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out, diameter_out, work, B, N,
                       only_if);
  } else {
    EML_ENSURE_LDS((&sinkhorn_loop_tiled_kernel<1>), lds);
    hipLaunchKernelGGL(sinkhorn_loop_tiled_kernel<1>, dim3(2 * B), dim3(1024), lds, stream, x, y, M, alpha, beta, blur,
                       log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out, diameter_out, work, B, N,
                       only_if);
  }
}

// Slices per problem pair for the split kernel, or 0 when the batch does not qualify.  S is sized so that every one of the
// 2 * B * S workgroups has a CU of its own among the CUs the STREAM may use (its CU mask / ROC_GLOBAL_CU_MASK): that is what
// makes the split worth it, and it implies what the exchange needs (the S slices of a (sample, role) co-resident) with a
// wide margin; the occupancy the runtime reports for the instance guards the degenerate cases (0: cannot launch).
// EML_SINKHORN_FORCE_SPLIT (tests): skip the sizing and launch S = 8 (or 4) regardless, to exercise the on-device rescue.
int split_slices(int B, int N, int flags, hipStream_t stream) {
  const int dev_cus = device_cu_count();
  if (flags & EML_SINKHORN_FORCE_SPLIT) return (2 * B * 8 <= dev_cus || N > 256) ? 8 : 4;
  const int cus = stream_cu_count(stream);
  int S = 0;
  if (2 * B * 8 <= cus) S = 8;
  else if (N <= 256 && 2 * B * 4 <= cus) S = 4;
  if (S == 0) return 0;
  const int R = N / S;
  const size_t lds = split_lds_bytes(N, R);
  int occ = 0;
#define EML_SPLIT_OCC(CPLV, RV)                                                                 \
  do {                                                                                          \
    static std::atomic<int> cache_[eml::kMaxDevices];                                           \
    EML_ENSURE_LDS((&sinkhorn_loop_split_kernel<CPLV, RV>), lds);                               \
    occ = split_occupancy(sinkhorn_loop_split_kernel<CPLV, RV>, 16 * RV, lds, cache_);          \
  } while (0)
  if (N == 256 && S == 8) EML_SPLIT_OCC(32, 32);
  else if (N == 256) EML_SPLIT_OCC(32, 64);
  else if (N == 192 && S == 8) EML_SPLIT_OCC(24, 24);
  else if (N == 192) EML_SPLIT_OCC(24, 48);
  else if (N == 320) EML_SPLIT_OCC(40, 40);
  else if (N == 384) EML_SPLIT_OCC(48, 48);
  else if (N == 448) EML_SPLIT_OCC(56, 56);
  else EML_SPLIT_OCC(64, 64);
#undef EML_SPLIT_OCC
  return ((long)2 * B * S <= (long)occ * cus) ? S : 0;
}
}  // namespace

extern "C" int eml_sinkhorn_fwd_f32(const float* x, const float* y, const float* M, const float* Mt,
                                    const float* alpha, const float* beta, double blur, double scaling, int p,
                                    double diameter, const float* range_lo_hi, float* eps_out, int* n_eps_out,
                                    float* diameter_out, float* loss, float* gx, float* gy, float* work, int B, int N,
                                    eml_stream_t stream) {
  return eml_sinkhorn_fwd_ex_f32(x, y, M, Mt, alpha, beta, blur, scaling, p, diameter, range_lo_hi, eps_out, n_eps_out,
                                 diameter_out, loss, gx, gy, work, B, N, 0, stream);
}

extern "C" int eml_sinkhorn_fwd_ex_f32(const float* x, const float* y, const float* M, const float* Mt,
                                       const float* alpha, const float* beta, double blur, double scaling, int p,
                                       double diameter, const float* range_lo_hi, float* eps_out, int* n_eps_out,
                                       float* diameter_out, float* loss, float* gx, float* gy, float* work, int B, int N,
                                       int flags, eml_stream_t stream) {
  if (!x || !y || !M || !Mt || !loss || !work) return eml::fail(EML_EINVAL, "eml_sinkhorn_fwd_f32: null pointer");
  if (flags & ~(EML_SINKHORN_NO_SPLIT | EML_SINKHORN_FORCE_SPLIT | EML_SINKHORN_TEST_STALL))
    return eml::fail(EML_EINVAL, "eml_sinkhorn_fwd_ex_f32: unknown flags 0x%x", flags);
  if (B < 0 || N < 1 || N > 2048) return eml::fail(EML_EINVAL, "eml_sinkhorn_fwd_f32: need 1<=N<=2048 (got %d)", N);
  if (!(blur > 0.0) || !(scaling > 0.0 && scaling < 1.0) || p < 1)
    return eml::fail(EML_EINVAL, "eml_sinkhorn_fwd_f32: need blur>0, 0<scaling<1, p>=1");
  if (B == 0) return EML_OK;
  const double log_blur = std::log(blur), log_scaling = std::log(scaling);   // f64 like numpy; only log(diameter) is data
  const int NP = round_up4(N) + kJPT;
  size_t lds = (size_t)(kSmemVecs * NP) * sizeof(float);
  int split_s = 0;
  if (N <= 4 * kCJ) {
    lds += (size_t)(N * (round_up4(N) + 4) + kJPT) * sizeof(float);
    EML_ENSURE_LDS((&sinkhorn_loop_kernel<true>), lds);
    hipLaunchKernelGGL(sinkhorn_loop_kernel<true>, dim3(2 * B), dim3(1024), lds, (hipStream_t)stream, x, y, M, Mt,
                       alpha, beta, blur, log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out, diameter_out,
                       work, B, N);
  } else if (N <= 512 && (N & 63) == 0 && N >= 192 && !(flags & EML_SINKHORN_NO_SPLIT) &&
             (split_s = split_slices(B, N, flags, (hipStream_t)stream)) > 0) {
    // small batch: the rows of every problem pair split over S workgroups (one per CU, all resident), duals exchanged
    // through global memory after every sweep (see the kernel).  S = 8 when 16 * B workgroups fit the CUs, else 4.
    const int S = split_s;
    const int R = N / S;
    unsigned long long* exch = reinterpret_cast<unsigned long long*>(work + (size_t)8 * B * N);
    int* status = reinterpret_cast<int*>(work + (size_t)24 * B * N);
    // one memset: the exchange granules and the status words behind them
    hipError_t me = hipMemsetAsync(exch, 0, ((size_t)16 * B * N + 4) * sizeof(float), (hipStream_t)stream);
    if (me != hipSuccess) return eml::fail(EML_ELAUNCH, "eml_sinkhorn_fwd_f32: memset of the exchange buffer: %s", hipGetErrorString(me));
    lds = split_lds_bytes(N, R);
#define EML_LAUNCH_SPLIT(CPLV, RV)                                                                                     \
  do {                                                                                                                 \
    EML_ENSURE_LDS((&sinkhorn_loop_split_kernel<CPLV, RV>), lds);                                                      \
    hipLaunchKernelGGL((sinkhorn_loop_split_kernel<CPLV, RV>), dim3(2 * B * S), dim3(16 * RV), lds, (hipStream_t)stream, \
                       x, y, M, alpha, beta, blur, log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out, \
                       diameter_out, work, exch, B, S, status, (flags & EML_SINKHORN_TEST_STALL) ? 1 : 0);              \
  } while (0)
    if (N == 256 && S == 8) EML_LAUNCH_SPLIT(32, 32);
    else if (N == 256) EML_LAUNCH_SPLIT(32, 64);
    else if (N == 192 && S == 8) EML_LAUNCH_SPLIT(24, 24);
    else if (N == 192) EML_LAUNCH_SPLIT(24, 48);
    else if (N == 320) EML_LAUNCH_SPLIT(40, 40);
    else if (N == 384) EML_LAUNCH_SPLIT(48, 48);
    else if (N == 448) EML_LAUNCH_SPLIT(56, 56);
    else EML_LAUNCH_SPLIT(64, 64);
#undef EML_LAUNCH_SPLIT
    int rcs = eml::check_launch("eml_sinkhorn_fwd_f32(split loop)");
    if (rcs) return rcs;
    // the rescue: the tiled kernel, gated on the status word -- returns at once unless a slice gave up
    launch_tiled(x, y, M, alpha, beta, blur, log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out,
                 diameter_out, work, B, N, status, (hipStream_t)stream);
  } else if (N <= 512 && (N & 3) == 0) {
    launch_tiled(x, y, M, alpha, beta, blur, log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out,
                 diameter_out, work, B, N, nullptr, (hipStream_t)stream);
  } else {
    EML_ENSURE_LDS((&sinkhorn_loop_kernel<false>), lds);
    hipLaunchKernelGGL(sinkhorn_loop_kernel<false>, dim3(2 * B), dim3(512), lds, (hipStream_t)stream, x, y, M, Mt,
                       alpha, beta, blur, log_blur, log_scaling, p, diameter, range_lo_hi, eps_out, n_eps_out, diameter_out,
                       work, B, N);
  }
  int rc = eml::check_launch("eml_sinkhorn_fwd_f32(loop)");
  if (rc) return rc;
  hipLaunchKernelGGL(sinkhorn_finish_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, work, alpha, beta, loss,
                     gx, gy, B, N);
  return eml::check_launch("eml_sinkhorn_fwd_f32(finish)");
}

extern "C" int eml_sinkhorn_bwd_f32(const float* gloss, const float* gunit, float* gout, int B, int N,
                                    eml_stream_t stream) {
  if (!gloss || !gunit || !gout || B < 0 || N < 1) return eml::fail(EML_EINVAL, "eml_sinkhorn_bwd_f32: bad arguments");
  if (B == 0) return EML_OK;
  const size_t total = (size_t)B * N;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     gloss, gunit, gout, B, N);
  return eml::check_launch("eml_sinkhorn_bwd_f32");
}
