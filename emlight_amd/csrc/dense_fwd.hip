// Forward kernels of the DenseNet-BC encoder (reference RegressionNetwork/DenseNet.py:14-65,
// 88-122) for gfx950, f32 MFMA (v_mfma_f32_16x16x4_f32: exact f32, 157 TF/s peak).
//
// Data layout: every dense block lives in ONE pixel-major (NHWC) buffer X[P][ld] whose
// channel axis is the concatenation, so `torch.cat` (DenseNet.py:55) is a pointer offset: a
// layer reads channels [0, C_in) and appends its 12 new channels at [C_in, C_in+12).
//
// conv1x1 (bottleneck, and the transitions): out[p][o] = sum_k relu(s_k X[p][k] + t_k) W[k][o]
//   BN1 + ReLU are applied in registers on the way from HBM to the MFMA A operand -- the
//   normalised activation never exists in memory.  The A operand is loaded straight from
//   global memory as float4 (16 B/lane): MFMA's k index is only a summation index, so step t
//   of a 16-channel group takes lane (row r, kk) 's channel 16j+4kk+t and the weights are
//   pre-permuted to match ([Kp/16][4][48][4], one ds_read_b128 per B fragment).  No LDS
//   staging of activations, no barrier in the main loop.
//   Transitions: pool_act writes the 2x2 mean of relu(bn(x)) once (the pool commutes with the 1x1 conv)
//   and the same kernel runs on it with a unit BN; the POOL template path (pool folded into the
//   operand load, one re-read of X per 48-channel output chunk) is kept for the C ABI's pool = 1.
// conv3x3 (48 -> 12): BN2 (no ReLU, DenseNet.py:38-43) is applied while a 10x34x48 halo tile
//   is staged into LDS (zero padding applied AFTER BN, as F.conv2d pads the BN output); the tile is
//   double-buffered and the next one is staged from inside the MFMA stream of the current one;
//   the 9x48x16 weight fragments stay in registers for the whole persistent loop.
// Batch statistics for train-mode BN are emitted by every producer's epilogue as per-block
//   f64 partial (sum, sumsq) and finished by bn_prepare -- deterministic, no atomics.
#include "eml_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// component t of a float4 (t is a compile-time constant after unrolling).  MFMA loops run t OUTERMOST so
// that consecutive MFMAs hit different accumulators: back-to-back MFMAs on one accumulator pay the
// 40-cycle dependent latency instead of the 32-cycle issue interval (MI355X_MICROARCH.md).
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

__device__ __forceinline__ float4 bn_relu4(float4 x, float4 s, float4 t) {
  float4 r;
  r.x = fmaxf(fmaf(x.x, s.x, t.x), 0.f);
  r.y = fmaxf(fmaf(x.y, s.y, t.y), 0.f);
  r.z = fmaxf(fmaf(x.z, s.z, t.z), 0.f);
  r.w = fmaxf(fmaf(x.w, s.w, t.w), 0.f);
  return r;
}

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// Reduce per-lane f64 (sum, sumsq) of NCH channels-per-lane over the 4 row groups of a wave
// and the 4 waves of a block; channel of (n, lane) is 16n + (lane & 15).
template <int NT>
__device__ __forceinline__ void block_stats_store(double (&s)[NT], double (&q)[NT], double* red /*LDS [4][NT*16][2]*/,
                                                  double* __restrict__ partials /*[NT*16][2]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    s[n] += shfl_xor_d(s[n], 16);
    s[n] += shfl_xor_d(s[n], 32);
    q[n] += shfl_xor_d(q[n], 16);
    q[n] += shfl_xor_d(q[n], 32);
    if (lane < 16) {
      red[(wave * NT * 16 + n * 16 + lane) * 2 + 0] = s[n];
      red[(wave * NT * 16 + n * 16 + lane) * 2 + 1] = q[n];
    }
  }
  __syncthreads();
  for (int e = tid; e < NT * 16 * 2; e += 256)
    partials[e] = (red[e] + red[NT * 32 + e]) + (red[2 * NT * 32 + e] + red[3 * NT * 32 + e]);
}

// ------------------------------------------------------------------------------ conv1x1
// out[p][n0 + o] (o < n_valid <= 48) for output pixels p < P.  POOL: the A operand is the
// 2x2 average of relu(bn(x)) (avg-pool commutes with the 1x1 conv: transition, DenseNet.py:14-21).
#ifndef EML_FWD_MIN_WG   // experiment builds only (tools/exp_build.sh): workgroups per CU the register allocation is capped for
#define EML_FWD_MIN_WG 2
#endif
template <bool POOL, bool MASK = false /* emit the ReLU ballot words (relu_mask != NULL, dense layers in training) */>
__global__ __launch_bounds__(256, EML_FWD_MIN_WG) void conv1x1_fwd_kernel(
    const float* __restrict__ X, int ldx, int P, int Hin, int Win, int Kp,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ Wp,
    float* __restrict__ out, int ldo, int n_valid, double* __restrict__ partials,
    unsigned long long* __restrict__ relu_mask, int staged) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                 // [Kp/16][4][48][4]
  float* sl = wl + (size_t)Kp * 48; // [Kp]
  float* tl = sl + Kp;              // [Kp]
  double* red = reinterpret_cast<double*>(tl + Kp);  // [4][48][2]
  // ReLU ballot words of the tile being computed, staged per wave: [4 waves][4 pixel groups][Kp/16][4] (round 3).  A
  // global store per K-step sat in the middle of the operand-load stream: vmcnt counts in order, so every step's wait
  // for its x operands (the ISA shows s_waitcnt vmcnt(0) at the top of the step) also waited for the PREVIOUS step's mask
  // store to reach memory.  Through LDS (lgkmcnt) the loop holds loads only; the words leave once per tile, as one
  // contiguous 128*Kp/16-byte run per wave.
  unsigned long long* mask_l = reinterpret_cast<unsigned long long*>(red + 4 * 48 * 2);
  // staged (round 6; dense layers, compact 48-wide output): a 16-pixel group's 16 x 48 results go through a wave-private LDS
  // tile (row stride 52 floats: conflict-free 16-byte writes) and leave as three stores of 1 KB of CONSECUTIVE addresses.  The
  // D^T epilogue stores 16 x 64 bytes per instruction -- half lines, which a write-only stream pays for: 3.3 TB/s against
  // 5.5 for whole lines on this very shape (profiles/r06_write_pattern.txt).
  float* stg = reinterpret_cast<float*>(mask_l + (MASK ? (size_t)4 * 16 * (Kp >> 4) : 0)) + (size_t)(threadIdx.x >> 6) * 16 * 52;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;

#pragma unroll 8
  for (int e = tid; e < Kp * 12; e += 256)
    reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(Wp)[e];
  for (int e = tid; e < Kp; e += 256) {
    sl[e] = scale[e];
    tl[e] = shift[e];
  }
#ifdef EML_FWD_STATS_LDS   // experiment build: the 24 f64 statistics accumulators in wave-private LDS slots (48 registers less)
  for (int e = tid; e < 4 * 48 * 2; e += 256) red[e] = 0.0;
#endif
  __syncthreads();

  // D^T form (weights as the MFMA A operand): lane (r, kk) owns output channels 16n + 4kk .. +3 of pixel
  // p0 + 16m + r -> 16-byte stores; per-lane statistics of those 12 channels.
#ifndef EML_FWD_STATS_LDS
  double ssum[3][4], ssq[3][4];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) ssum[n][g] = ssq[n][g] = 0.0;
#endif
  const int nj = Kp >> 4;
  const int ntiles = (P + 255) >> 8;
  const int Wo = Win >> 1, Ho = Hin >> 1;
  const bool vec_ok = (ldo & 3) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0;

  auto row_ptr = [&](int tile, int m) -> const float* {
    const int pm = min(tile * 256 + wave * 64 + 16 * m + r, P - 1);
    if constexpr (POOL) {
      const int b = pm / (Ho * Wo), rem = pm - b * (Ho * Wo);
      const int oy = rem / Wo, ox = rem - oy * Wo;
      return X + ((size_t)(b * Hin + 2 * oy) * Win + 2 * ox) * ldx + 4 * kk;
    } else {
      return X + (size_t)pm * ldx + 4 * kk;
    }
  };
  // Dense layers (!POOL): the raw x of K-step j+1 -- or of the next tile's first K-step -- is requested before the
  // MFMAs of K-step j (unconditional loads, clamped rows).  Issued inside its own K-step, every 16-channel step
  // exposed an HBM round trip (ISA: global_load; s_waitcnt vmcnt; v_mfma).
  float4 xn[4];
  int tile = blockIdx.x;
  if constexpr (!POOL) {
    if (tile < ntiles) {
#pragma unroll
      for (int m = 0; m < 4; ++m) xn[m] = *reinterpret_cast<const float4*>(row_ptr(tile, m));
    }
  }
  for (; tile < ntiles; tile += gridDim.x) {
    const int p0 = tile * 256 + wave * 64;
    const float* rp[4];
    const float* rpn[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      rp[m] = row_ptr(tile, m);
      if constexpr (!POOL) rpn[m] = row_ptr(min(tile + (int)gridDim.x, ntiles - 1), m);
    }
    f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](int j) {
      const float4 s4 = *reinterpret_cast<const float4*>(sl + 16 * j + 4 * kk);
      const float4 t4 = *reinterpret_cast<const float4*>(tl + 16 * j + 4 * kk);
      float4 a[4];
      if constexpr (POOL) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float* q = rp[m] + 16 * j;
          const float4 v0 = bn_relu4(*reinterpret_cast<const float4*>(q), s4, t4);
          const float4 v1 = bn_relu4(*reinterpret_cast<const float4*>(q + ldx), s4, t4);
          const float4 v2 = bn_relu4(*reinterpret_cast<const float4*>(q + (size_t)Win * ldx), s4, t4);
          const float4 v3 = bn_relu4(*reinterpret_cast<const float4*>(q + (size_t)Win * ldx + ldx), s4, t4);
          a[m].x = ((v0.x + v1.x) + (v2.x + v3.x)) * 0.25f;
          a[m].y = ((v0.y + v1.y) + (v2.y + v3.y)) * 0.25f;
          a[m].z = ((v0.z + v1.z) + (v2.z + v3.z)) * 0.25f;
          a[m].w = ((v0.w + v1.w) + (v2.w + v3.w)) * 0.25f;
        }
      } else {
        float4 xc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xc[m] = xn[m];
        const bool last = j + 1 == nj;  // wave-uniform
#pragma unroll
        for (int m = 0; m < 4; ++m) xn[m] = *reinterpret_cast<const float4*>(last ? rpn[m] : rp[m] + 16 * (j + 1));
        __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks these requests below the MFMAs
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = bn_relu4(xc[m], s4, t4);
#ifndef EML_FWD_NOMASK   // experiment build (tools/exp_build.sh nomask -DEML_FWD_NOMASK): what the ballots + 16 selects + the store cost
        if constexpr (MASK) {
          // ReLU mask of this K-step as wave ballots: word (pixel group, K-step, t), bit r + 16*kk <-> pixel 16*pg + r,
          // channel 16*j + 4*kk + t -- the lane layout of the data-gradient kernel, which then needs neither X nor
          // BN1's affine to know where relu(bn1(x)) was active.  Lane i < 16 stores word (m = i / 4, t = i % 4).
          // (tried: v_writelane of each ballot into its lane via inline asm -- no faster, and inline asm gets no hazard
          // handling after the v_cmp that writes the SGPR pair: wrong bits)
          unsigned long long mine = 0ull;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const unsigned long long w = __ballot(f4c(a[m], t) > 0.f);
              mine = (lane == 4 * m + t) ? w : mine;
            }
          if (lane < 16) mask_l[((wave * 4 + (lane >> 2)) * nj + j) * 4 + (lane & 3)] = mine;
        }
#endif
      }
      float4 bw[3];
#pragma unroll
      for (int n = 0; n < 3; ++n)
        bw[n] = *reinterpret_cast<const float4*>(wl + ((size_t)(j * 4 + kk) * 48 + 16 * n + r) * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[m][n] = mfma16(f4c(bw[n], t), f4c(a[m], t), acc[m][n]);
    };
    if constexpr (POOL) {
      for (int j = 0; j < nj; ++j) kstep(j);
    } else {
#pragma unroll 2
      for (int j = 0; j < nj; ++j) kstep(j);
    }
    if constexpr (!POOL) {
      if constexpr (MASK) {   // this wave's words of the tile: one contiguous run (wave-private LDS, in-order: compiler fence only)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long* dst = relu_mask + (size_t)(tile * 16 + wave * 4) * nj * 4;
        for (int e = lane; e < 16 * nj; e += 64) dst[e] = mask_l[wave * 16 * nj + e];
      }
    }
    // epilogue: D rows = channels 16n + 4kk + g, column = pixel 16m + r
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int c4 = 16 * n + 4 * kk;
      float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int p = p0 + 16 * m + r;
        const bool pv = p < P;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v = (pv && c4 + g < n_valid) ? acc[m][n][g] : 0.f;
          ls[g] += v;
          lq[g] = fmaf(v, v, lq[g]);
        }
        if (pv && !staged) {
          float* dst = out + (size_t)p * ldo + c4;
          if (vec_ok && c4 + 4 <= n_valid) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
          } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              if (c4 + g < n_valid) dst[g] = acc[m][n][g];
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#ifdef EML_FWD_STATS_LDS
        const float t1 = eml::row16_sum(ls[g]), t2 = eml::row16_sum(lq[g]);
        if (r == 0) {
          double* d = red + (wave * 48 + 16 * n + 4 * kk + g) * 2;
          d[0] += (double)t1;
          d[1] += (double)t2;
        }
#else
        ssum[n][g] += (double)ls[g];
        ssq[n][g] += (double)lq[g];
#endif
      }
    }
    if (staged) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int n = 0; n < 3; ++n) *reinterpret_cast<f32x4*>(stg + r * 52 + 16 * n + 4 * kk) = acc[m][n];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private tile, LDS runs a wave's accesses in order
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int pg = p0 + 16 * m;
        float* dst = out + (size_t)pg * 48;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int f = lane + 64 * i, px = f / 12, piece = f - 12 * px;
          const f32x4 v = *reinterpret_cast<const f32x4*>(stg + px * 52 + 4 * piece);
          if (pg + px < P) *reinterpret_cast<f32x4*>(dst + 4 * f) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  // channel statistics: over the 16 pixel lanes r, then the 4 waves
#ifndef EML_FWD_STATS_LDS
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        ssum[n][g] += shfl_xor_d(ssum[n][g], o);
        ssq[n][g] += shfl_xor_d(ssq[n][g], o);
      }
      if (r == 0) {
        red[(wave * 48 + 16 * n + 4 * kk + g) * 2 + 0] = ssum[n][g];
        red[(wave * 48 + 16 * n + 4 * kk + g) * 2 + 1] = ssq[n][g];
      }
    }
#endif
  __syncthreads();
  for (int e = tid; e < 96; e += 256)
    partials[(size_t)blockIdx.x * 96 + e] = (red[e] + red[96 + e]) + (red[192 + e] + red[288 + e]);
}

// ------------------------------------------------------------------------------ conv3x3
constexpr int kTH = 8, kTW = 32;            // output tile per block
constexpr int kHH = kTH + 2, kHW = kTW + 2; // halo tile
constexpr int kPS = 56;                     // LDS pixel stride: 56 dwords makes the ds_read_b128 lane groups conflict-free (48 and 52 are 2-way)

// 512 threads = 8 waves, ONE output row of the 8x32 tile per wave, one workgroup per CU.  The halo tile is
// double-buffered in LDS (2 x 70.7 KB): the global loads of tile t+1 are issued into registers before the
// MFMAs of tile t and committed (BN2 applied) to the other buffer after them, so a tile costs one barrier and
// the HBM round trip hides behind 54 MFMAs/wave x 2 waves/SIMD.  (The single-buffered 256-thread version
// alternated "all waves wait for HBM" / "all waves compute": 39 % of the f32 MFMA peak, 1.9 TB/s.)
constexpr int kC3Threads = 512;

__global__ __launch_bounds__(kC3Threads) void conv3x3_fwd_kernel(
    const float* __restrict__ Z, const float* __restrict__ scale2, const float* __restrict__ shift2,
    const float* __restrict__ W2p, float* __restrict__ X, int ldx, int c_out0, int B, int H, int W,
    double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* st_l = smem + 2 * kHH * kHW * kPS;               // [2][48]
  double* red = reinterpret_cast<double*>(st_l + 96);     // [8][16][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;

  if (tid < 48) {
    st_l[tid] = scale2[tid];
    st_l[48 + tid] = shift2[tid];
  }
  // weight fragments: lane (kk, o=r) holds W2[o][16j+4kk+t][tap], t = 0..3
  float4 bw[9][3];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      bw[tap][j] = *reinterpret_cast<const float4*>(W2p + ((size_t)((tap * 3 + j) * 4 + kk) * 16 + r) * 4);
  __syncthreads();

  const int tx_n = (W + kTW - 1) / kTW, ty_n = (H + kTH - 1) / kTH;
  const int ntiles = B * ty_n * tx_n;
  const bool aligned16 = ((c_out0 | ldx) & 3) == 0;  // block-uniform
  double ssum[4] = {0.0, 0.0, 0.0, 0.0}, ssq[4] = {0.0, 0.0, 0.0, 0.0};

  // staging map: threads 0..407 = (halo column hx = tid/12, 16-byte slice q = tid%12), one halo ROW per pass:
  // the index math is tile-invariant and every LDS / global offset of pass `it` is base + it*const.
  // Loads are UNCONDITIONAL (coordinates clamped into the image, threads >= 408 re-read column 33) and the
  // in-image mask is applied at commit: exec-masked loads sit in their own basic blocks and the compiler then
  // makes MFMAs wait on them.
  const int s_hx = min(tid / 12, kHW - 1), s_q = tid % 12;
  const float4 sc = *reinterpret_cast<const float4*>(scale2 + 4 * s_q);
  const float4 sh = *reinterpret_cast<const float4*>(shift2 + 4 * s_q);
  const int s_dst = s_hx * kPS + 4 * s_q;
  float4 zt[kHH];
  // state of the tile being staged
  const float* s_src = Z;
  int s_y0 = 0;
  bool s_col = false;
  auto stage_begin = [&](int tile) {
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
    const int gx = tx * kTW - 1 + s_hx;
    s_y0 = ty * kTH - 1;
    s_col = gx >= 0 && gx < W;
    s_src = Z + ((size_t)b * H * W + min(max(gx, 0), W - 1)) * 48 + 4 * s_q;
  };
  auto stage_load = [&](int it) {
    zt[it] = *reinterpret_cast<const float4*>(s_src + (size_t)min(max(s_y0 + it, 0), H - 1) * W * 48);
  };
  // BN2 + zero padding of the BN OUTPUT (out-of-image halo pixels are 0, not shift2)
  auto stage_commit = [&](int it, float* tile_l) {
    const bool ok = s_col && s_y0 + it >= 0 && s_y0 + it < H;
    float4 v;
    v.x = ok ? fmaf(zt[it].x, sc.x, sh.x) : 0.f;
    v.y = ok ? fmaf(zt[it].y, sc.y, sh.y) : 0.f;
    v.z = ok ? fmaf(zt[it].z, sc.z, sh.z) : 0.f;
    v.w = ok ? fmaf(zt[it].w, sc.w, sh.w) : 0.f;
    *reinterpret_cast<float4*>(tile_l + s_dst + it * kHW * kPS) = v;  // threads >= 408 duplicate column 33: same value
  };

  int tile = blockIdx.x, cur = 0;
  if (tile < ntiles) {
    stage_begin(tile);
#pragma unroll
    for (int it = 0; it < kHH; ++it) stage_load(it);
#pragma unroll
    for (int it = 0; it < kHH; ++it) stage_commit(it, smem);
  }
  // vmcnt(0): the weight-fragment loads above are complete on EVERY path into the loop.  Without it the
  // compiler's waitcnt pass keeps bw[] "possibly pending" and guards the first MFMAs of each tile with
  // s_waitcnt vmcnt(3..0) -- which at run time waits for the prefetch loads issued just before them.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    // The next tile is staged INSIDE this tile's MFMA stream: its 10 row loads ride on the first 10 of the 27
    // (tap, 16-channel) groups, its 10 LDS commits on the last 10 -- the two waves of a SIMD run in lockstep
    // between barriers, so any staging done outside the MFMA stream leaves the matrix pipe idle.  On the last
    // tile the block re-stages the same tile (harmless) to keep the stream branch-free.
    const int nxt = tile + gridDim.x;
    stage_begin(nxt < ntiles ? nxt : tile);
    const float* tile_l = smem + cur * (kHH * kHW * kPS);
    float* tile_n = smem + (cur ^ 1) * (kHH * kHW * kPS);
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;

    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // wave w owns output row w; pixel tile m = 16 columns at 16*m.  D^T form (weights as the A operand): lane
    // (r, kk) ends up with output channels 4kk..4kk+3 of pixel 16m + r -> one 16-byte store per pixel tile.
    float4 aq[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) aq[0][m] = *reinterpret_cast<const float4*>(tile_l + (wave * kHW + 16 * m + r) * kPS + 4 * kk);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int gi = tap * 3 + j;
        if (gi < kHH) {
          stage_load(gi);
          __builtin_amdgcn_sched_barrier(0);  // keep the load HERE: the scheduler otherwise sinks it to its use
        }
        // the operand of group gi + 1 is requested before the MFMAs of group gi (round 4: read right in front of its
        // MFMAs, every group of 8 waited out an LDS round trip -- both waves of a SIMD share the phase here)
        if (gi + 1 < 27) {
          const int tn = (gi + 1) / 3, jn = (gi + 1) - 3 * tn, dyn = tn / 3, dxn = tn - 3 * dyn;
#pragma unroll
          for (int m = 0; m < 2; ++m)
            aq[(gi + 1) & 1][m] =
                *reinterpret_cast<const float4*>(tile_l + ((wave + dyn) * kHW + 16 * m + r + dxn) * kPS + 16 * jn + 4 * kk);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma16(f4c(bw[tap][j], t), f4c(aq[gi & 1][m], t), acc[m]);
        if (gi >= 27 - kHH) {
          __builtin_amdgcn_sched_barrier(0);
          stage_commit(gi - (27 - kHH), tile_n);
        }
      }
    }
    const int gy = ty * kTH + wave;
    float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int gx = tx * kTW + 16 * m + r;
      const bool ok = gy < H && gx < W && kk < 3;  // channels 12..15 are MFMA padding
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v = ok ? acc[m][g] : 0.f;
        ls[g] += v;
        lq[g] = fmaf(v, v, lq[g]);
      }
      if (ok) {
        float* dst = X + ((size_t)(b * H + gy) * W + gx) * ldx + c_out0 + 4 * kk;
        if (aligned16) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        } else {  // third dense block: c_out0 = 150 + 12 l is only 8-byte aligned
          *reinterpret_cast<float2*>(dst) = make_float2(acc[m][0], acc[m][1]);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(acc[m][2], acc[m][3]);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ssum[g] += (double)ls[g];
      ssq[g] += (double)lq[g];
    }
    eml::lds_barrier();  // buffer cur^1 is complete; everyone is done reading buffer cur (LDS only: no wait for the stores)
    cur ^= 1;
  }
  // channel statistics of the 12 new channels: channel 4kk+g over the 16 pixel lanes r, then the 8 waves
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      ssum[g] += shfl_xor_d(ssum[g], o);
      ssq[g] += shfl_xor_d(ssq[g], o);
    }
    if (r == 0) {
      red[(wave * 16 + 4 * kk + g) * 2 + 0] = ssum[g];
      red[(wave * 16 + 4 * kk + g) * 2 + 1] = ssq[g];
    }
  }
  __syncthreads();
  if (tid < 32) {
    double t = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) t += red[w8 * 32 + tid];
    partials[(size_t)blockIdx.x * 32 + tid] = t;
  }
}

// ------------------------------------------------------------------------------ conv0 (3 -> C0, 3x3, NCHW in)
template <int C0>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                        float* __restrict__ X, int ldx, int B, int H, int W,
                                                        double* __restrict__ partials) {
  __shared__ float wl[C0 * 27];
  __shared__ double wacc[4][C0][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < C0 * 27; e += 256) wl[e] = w0[e];  // [o][c][ky][kx]
  for (int e = tid; e < 4 * C0 * 2; e += 256) (&wacc[0][0][0])[e] = 0.0;
  __syncthreads();
  const size_t P = (size_t)B * H * W, plane = (size_t)H * W;
  const size_t nt = (P + 255) / 256;
  for (size_t tile = blockIdx.x; tile < nt; tile += gridDim.x) {
    const size_t p = tile * 256 + tid;
    const bool valid = p < P;
    const size_t pc = valid ? p : P - 1;
    const int b = (int)(pc / plane);
    const int rem = (int)(pc - (size_t)b * plane);
    const int yy = rem / W, xx = rem - yy * W;
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int gy = yy + ky - 1, gx = xx + kx - 1;
          in[c * 9 + ky * 3 + kx] =
              (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)(b * 3 + c) * H + gy) * W + gx] : 0.f;
        }
    // 8 output channels at a time keeps the register footprint small (27 inputs + 8 sums)
#pragma unroll 1
    for (int g8 = 0; g8 < C0 / 8; ++g8) {
      float o[8];
#pragma unroll
      for (int oc = 0; oc < 8; ++oc) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) s = fmaf(in[k], wl[(g8 * 8 + oc) * 27 + k], s);
        o[oc] = s;
      }
      if (valid) {
        float4* dst = reinterpret_cast<float4*>(X + p * ldx + g8 * 8);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
      }
#pragma unroll
      for (int oc = 0; oc < 8; ++oc) {
        const float v = valid ? o[oc] : 0.f;
        const float s = eml::wave_sum(v), q = eml::wave_sum(v * v);
        if (lane == 0) {
          wacc[wave][g8 * 8 + oc][0] += (double)s;
          wacc[wave][g8 * 8 + oc][1] += (double)q;
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < C0 * 2; e += 256) {
    const int oc = e >> 1, k = e & 1;
    partials[(size_t)blockIdx.x * C0 * 2 + e] = (wacc[0][oc][k] + wacc[1][oc][k]) + (wacc[2][oc][k] + wacc[3][oc][k]);
  }
}

// conv0 on the matrix unit (round 6; the engine's default): D[o][p] = sum_k W0[o][k] in[k][p], k = (c, ky, kx) -- 27 of 28 -- in the D^T
// form of the other kernels: A = W0 (two 16-row tiles: outputs 0..15 and 16..23), B = the 3x3x3 neighbourhood of 16 pixels,
// lane (r, kk) supplying element k = 4 s + kk of pixel r at step s -- ONE 4-byte load per lane, step and 16 pixels (7 per 16
// pixels instead of 27 per pixel), 14 MFMAs per 16 pixels instead of 648 FMAs per pixel each with its LDS weight operand
// (the VALU kernel above is bound by those broadcast reads: 0.51 ms at 64 x 240 x 320).  A lane then holds channels
// 4kk..4kk+3 (and 16+4kk.., kk < 2) of its pixel: 16-byte stores into the compact (P, 24) rows, BatchNorm sums by DPP row
// reductions.  Bit for bit the VALU kernel's outputs and partial sums (v_mfma_f32_16x16x4_f32 adds its k terms in order, like
// that kernel's fma chain over k = 0..26; the sums below follow its wave_sum's association): a first version whose sums were
// merely as accurate moved the golden train step's sampled gradients from 0.011 to 0.033 of a tensor's RMS.
__global__ __launch_bounds__(256) void conv0_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                             float* __restrict__ X, int ldx, int B, int H, int W,
                                                             double* __restrict__ partials) {
  constexpr int C0 = 24, MT = 4;
  __shared__ double wacc[4][C0][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  for (int e = tid; e < 4 * C0 * 2; e += 256) (&wacc[0][0][0])[e] = 0.0;
  // weights: A[row r of tile j][k = 4 s + kk]
  float wa[2][7];
  int off[7], dy[7], dx[7];
  const int plane = H * W;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + kk;
    const bool kin = k < 27;
    const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
    wa[0][s] = kin ? w0[r * 27 + k] : 0.f;
    wa[1][s] = (kin && r < 8) ? w0[(16 + r) * 27 + k] : 0.f;
    dy[s] = kin ? ky - 1 : 0;
    dx[s] = kin ? kx - 1 : 0;
    off[s] = kin ? c * plane + (ky - 1) * W + (kx - 1) : 0;
    if (!kin) dy[s] = 1 << 20;   // never inside: the padded k contributes 0
  }
  __syncthreads();
  const long P = (long)B * plane;
  const long nt = (P + 64 * 4 - 1) / (64 * 4);   // a workgroup tile = 4 waves x 64 pixels
  float ls[8], lq[8];
  for (long tile = blockIdx.x; tile < nt; tile += gridDim.x) {
    const long p0 = tile * 256 + wave * 64;
    float bv[MT][7];
    long pp[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const long p = p0 + 16 * m + r;
      pv[m] = p < P;
      pp[m] = pv[m] ? p : P - 1;
      const int b = (int)(pp[m] / plane);
      const int rem = (int)(pp[m] - (long)b * plane);
      const int yy = rem / W, xx = rem - yy * W;
      const float* base = x + (long)b * 3 * plane + rem;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const int gy = yy + dy[s], gx = xx + dx[s];
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        bv[m][s] = in ? base[off[s]] : 0.f;
      }
    }
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][j] = mfma16(wa[j][s], bv[m][s], acc[m][j]);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (pv[m]) {
        *reinterpret_cast<f32x4*>(X + pp[m] * ldx + 4 * kk) = acc[m][0];
        if (kk < 2) *reinterpret_cast<f32x4*>(X + pp[m] * ldx + 16 + 4 * kk) = acc[m][1];
      }
    }
    // BatchNorm sums of the 64 pixels, in EXACTLY the association of the VALU kernel's wave_sum over lane = pixel (xor 32, 16,
    // 8, 4, 2, 1; squares rounded before they are added): pixel 16 m + r, so xor 32 / 16 pair the m's and the rest runs inside
    // the 16-lane row -- the partial sums, like the outputs (the MFMA adds its k terms in order), are bit for bit the same
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float v[MT], q[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        v[m] = pv[m] ? acc[m][g >> 2][g & 3] : 0.f;
        q[m] = v[m] * v[m];
      }
      ls[g] = (v[0] + v[2]) + (v[1] + v[3]);
      lq[g] = (q[0] + q[2]) + (q[1] + q[3]);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float t1 = ls[g], t2 = lq[g];
      t1 += eml::dpp_mov<0x128>(t1);   // row_ror:8 = lane ^ 8 inside the row
      t2 += eml::dpp_mov<0x128>(t2);
      t1 += __shfl_xor(t1, 4, 64);
      t2 += __shfl_xor(t2, 4, 64);
      t1 += eml::dpp_mov<0x4E>(t1);    // lane ^ 2
      t2 += eml::dpp_mov<0x4E>(t2);
      t1 += eml::dpp_mov<0xB1>(t1);    // lane ^ 1
      t2 += eml::dpp_mov<0xB1>(t2);
      const int ch = g < 4 ? 4 * kk + g : 16 + 4 * kk + (g - 4);
      if (r == 0 && (g < 4 || kk < 2)) {
        wacc[wave][ch][0] += (double)t1;
        wacc[wave][ch][1] += (double)t2;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < C0 * 2; e += 256) {
    const int oc = e >> 1, k = e & 1;
    partials[(size_t)blockIdx.x * C0 * 2 + e] = (wacc[0][oc][k] + wacc[1][oc][k]) + (wacc[2][oc][k] + wacc[3][oc][k]);
  }
}

// ------------------------------------------------------------------------------ BN apply (+ReLU) with stats of the result
// dst[p][c] = act(scale[c]*src[p][c] + shift[c]) for c < C; emits f64 partial stats of dst.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst,
                                                       int ldd, int C, size_t P, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu,
                                                       double* __restrict__ partials /*[G][C][2]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* red = reinterpret_cast<double*>(smem);  // [4][CP][2], CP = 64*ceil(C/64)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = (C + 63) >> 6;  // <= 6
  float s[6], t[6];
  double acc_s[6], acc_q[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int c = lane + 64 * k;
    s[k] = (k < nk && c < C) ? scale[c] : 0.f;
    t[k] = (k < nk && c < C) ? shift[c] : 0.f;
    acc_s[k] = acc_q[k] = 0.0;
  }
  // kUN pixels per wave and iteration: all their loads are issued (unconditionally, from clamped rows) before the
  // first is used -- one pixel at a time left a single 256-byte request per wave in flight (2 TB/s)
  constexpr int kUN = 4;
  for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * kUN; p0 < P; p0 += (size_t)gridDim.x * 4 * kUN) {
    float v[kUN][6];
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const size_t pu = (p0 + u < P) ? p0 + u : P - 1;
#pragma unroll
      for (int k = 0; k < 6; ++k) v[u][k] = src[pu * lds_ + min(lane + 64 * k, C - 1)];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int c = lane + 64 * k;
      if (k < nk && c < C) {
        // per-element f64 accumulation (batched loads, per-element converts): sum and sum of squares of 4.9 M pixels
#pragma unroll
        for (int u = 0; u < kUN; ++u)
          if (p0 + u < P) {
            float w = fmaf(v[u][k], s[k], t[k]);
            if (relu) w = fmaxf(w, 0.f);
            dst[(p0 + u) * ldd + c] = w;
            const double wd = (double)w;
            acc_s[k] += wd;
            acc_q[k] = fma(wd, wd, acc_q[k]);
          }
      }
    }
  }
  const int CP = nk * 64;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (k < nk) {
      red[(wave * CP + lane + 64 * k) * 2 + 0] = acc_s[k];
      red[(wave * CP + lane + 64 * k) * 2 + 1] = acc_q[k];
    }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256)
    partials[(size_t)blockIdx.x * C * 2 + e] =
        (red[e] + red[CP * 2 + e]) + (red[2 * CP * 2 + e] + red[3 * CP * 2 + e]);
}

// The same pass for FEW channels (C <= 32, a multiple of 4: norm0's 24 -- the only call at the full resolution).  The kernel
// above gives a lane a channel: with 24 channels 24 of 64 lanes work, every access is 4 bytes and a wave instruction moves 96
// bytes (and the six channel slots are loaded whether they exist or not): 4.9 M pixels were 14.7 M wave instructions, bound by
// the texture-address unit at 2.6 TB/s.  Here a lane owns 4 consecutive channels of a pixel (16-byte accesses), a wave
// instruction covers 64 / (C / 4) pixels, kUN of them in flight; the per-lane f64 sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void bn_apply_small_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst,
                                                             int ldd, int C, size_t P, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int relu,
                                                             double* __restrict__ partials /*[G][C][2]*/) {
  __shared__ double red[4][64][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CQ = C >> 2, PW = 64 / CQ;                 // channel quads per pixel, pixels per wave instruction
  const int slot = lane / CQ, quad = lane - slot * CQ;
  const bool live = slot < PW;
  const float4 s4 = *reinterpret_cast<const float4*>(scale + 4 * quad), t4 = *reinterpret_cast<const float4*>(shift + 4 * quad);
  double as[4] = {0.0, 0.0, 0.0, 0.0}, aq[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int kUN = 4;
  const size_t step = (size_t)gridDim.x * 4 * kUN * PW;
  for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * kUN * PW; p0 < P; p0 += step) {
    float4 v[kUN];
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const size_t p = p0 + (size_t)u * PW + slot;
      v[u] = *reinterpret_cast<const float4*>(src + (p < P ? p : P - 1) * lds_ + 4 * quad);
    }
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const size_t p = p0 + (size_t)u * PW + slot;
      if (live && p < P) {
        float w[4] = {fmaf(v[u].x, s4.x, t4.x), fmaf(v[u].y, s4.y, t4.y), fmaf(v[u].z, s4.z, t4.z), fmaf(v[u].w, s4.w, t4.w)};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (relu) w[g] = fmaxf(w[g], 0.f);
          const double wd = (double)w[g];
          as[g] += wd;
          aq[g] = fma(wd, wd, aq[g]);
        }
        *reinterpret_cast<float4*>(dst + p * ldd + 4 * quad) = make_float4(w[0], w[1], w[2], w[3]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    red[wave][lane][2 * g] = live ? as[g] : 0.0;
    red[wave][lane][2 * g + 1] = live ? aq[g] : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256) {
    const int c = e >> 1, k = e & 1, q = c >> 2, g = c & 3;
    double t = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int sl = 0; sl < PW; ++sl) t += red[w][sl * CQ + q][2 * g + k];
    partials[(size_t)blockIdx.x * C * 2 + e] = t;
  }
}

// ------------------------------------------------------------------------------ BN prepare
// (1) fold the f64 partial stats of n_new freshly produced channels [c_new0, c_new0+n_new)
//     into the block's per-channel mean / biased var / invstd arrays;
// (2) scale_k = gamma_k * invstd_k, shift_k = beta_k - mean_k * scale_k for k < C (zero up to Cpad);
// (3) training: running stats update with momentum (unbiased var), as nn.BatchNorm2d does.
// eval (training == 0): mean / var are taken from the running buffers.
__global__ __launch_bounds__(256) void bn_prepare_kernel(
    const double* __restrict__ partials, int G, int pstride, int n_new, int c_new0, double count,
    float* __restrict__ mean, float* __restrict__ var, float* __restrict__ istd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
    float* __restrict__ rvar, int C, int Cpad, float eps, float momentum, int training,
    float* __restrict__ scale, float* __restrict__ shift) {
  // one wavefront per channel: lanes stride over the G partial rows, f64 wave reduction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = blockIdx.x * 4 + wave;
  const int total = scale ? Cpad : n_new;
  if (idx >= total) return;
  const int c = scale ? idx : c_new0 + idx;
  const bool is_new = partials && c >= c_new0 && c < c_new0 + n_new;
  float m = 0.f, is = 0.f, v = 0.f;
  if (is_new) {
    double s = 0.0, q = 0.0;
    for (int g = lane; g < G; g += 64) {
      s += partials[(size_t)g * pstride + 2 * (c - c_new0)];
      q += partials[(size_t)g * pstride + 2 * (c - c_new0) + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    const double md = s / count;
    const double vd = fmax(q / count - md * md, 0.0);
    m = (float)md;
    v = (float)vd;
    is = (float)(1.0 / sqrt(vd + (double)eps));
    if (lane == 0) {
      mean[c] = m;
      var[c] = v;
      istd[c] = is;
    }
  } else if (scale && c < C && training) {
    m = mean[c];
    v = var[c];
    is = istd[c];
  }
  if (!scale || lane != 0) return;
  float sc = 0.f, sh = 0.f;
  if (c < C) {
    if (training) {
      const double unbiased = (double)v * (count / fmax(count - 1.0, 1.0));
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    } else {
      m = rmean[c];
      is = (float)(1.0 / sqrt((double)rvar[c] + (double)eps));
    }
    sc = gamma[c] * is;
    sh = beta[c] - m * sc;
  }
  scale[c] = sc;
  shift[c] = sh;
}

// ------------------------------------------------------------------------------ weight permutes
// W [Cout][Cin] (1x1 conv, PyTorch layout) -> Wp[chunk][Kp/16][4][48][4] with
// Wp[..][j][kk][o][t] = W[n0+o][16j+4kk+t], zero outside.
__global__ void permute_w1_kernel(const float* __restrict__ W, int Cout, int Cin, int Kp, int nchunks,
                                  float* __restrict__ Wp) {
  const size_t total = (size_t)nchunks * Kp * 48;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(e & 3);
    size_t rest = e >> 2;
    const int o = (int)(rest % 48);
    rest /= 48;
    const int kk = (int)(rest & 3);
    rest >>= 2;
    const int j = (int)(rest % (Kp >> 4));
    const int ch = (int)(rest / (Kp >> 4));
    const int k = 16 * j + 4 * kk + t, oc = ch * 48 + o;
    Wp[e] = (k < Cin && oc < Cout) ? W[(size_t)oc * Cin + k] : 0.f;
  }
}
// W2 [12][48][3][3] -> W2p[tap][j][kk][16][4] with W2p = W2[o][16j+4kk+t][tap], zero for o >= 12.
__global__ void permute_w2_kernel(const float* __restrict__ W2, int Cout, float* __restrict__ W2p) {
  const int total = 9 * 3 * 4 * 16 * 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int t = e & 3, o = (e >> 2) & 15, kk = (e >> 6) & 3;
    const int rest = e >> 8, j = rest % 3, tap = rest / 3;
    const int c = 16 * j + 4 * kk + t;
    W2p[e] = (o < Cout) ? W2[((size_t)o * 48 + c) * 9 + tap] : 0.f;
  }
}

// All weight re-layouts of a pass in ONE launch (blockIdx.y = descriptor): the per-layer permutes are 4-5 us kernels, 150 of
// them per training step, each one a launch boundary between two big kernels.  kind 0: permute_w1 (1x1 forward),
// 1: permute_w2 (3x3 forward), 2: the data-gradient layout of dense_bwd.hip's permute_w1_bwd_kernel, 3: permute_w2_tp.
__global__ __launch_bounds__(256) void permute_batch_kernel(const eml_permute_desc* __restrict__ descs) {
  const eml_permute_desc d = descs[blockIdx.y];
  const float* __restrict__ W = d.src;
  float* __restrict__ out = d.dst;
  const size_t stride = (size_t)gridDim.x * 256, e0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (d.kind == 0) {
    const int nj = d.Kp >> 4;
    const size_t total = (size_t)((d.Cout + 47) / 48) * d.Kp * 48;
    for (size_t e = e0; e < total; e += stride) {
      const int t = (int)(e & 3);
      size_t rest = e >> 2;
      const int o = (int)(rest % 48);
      rest /= 48;
      const int kk = (int)(rest & 3);
      rest >>= 2;
      const int j = (int)(rest % nj), ch = (int)(rest / nj);
      const int k = 16 * j + 4 * kk + t, oc = ch * 48 + o;
      out[e] = (k < d.Cin && oc < d.Cout) ? W[(size_t)oc * d.Cin + k] : 0.f;
    }
  } else if (d.kind == 1) {
    for (size_t e = e0; e < 9 * 3 * 4 * 16 * 4; e += stride) {
      const int ei = (int)e, t = ei & 3, o = (ei >> 2) & 15, kk = (ei >> 6) & 3;
      const int rest = ei >> 8, j = rest % 3, tap = rest / 3;
      out[e] = (o < d.Cout) ? W[((size_t)o * 48 + 16 * j + 4 * kk + t) * 9 + tap] : 0.f;
    }
  } else if (d.kind == 3) {   // tap-packed conv3x3 weights (dense_fwd_tp.hip)
    for (size_t e = e0; e < (size_t)eml::kW2tFloats; e += stride) out[e] = eml::w2t_value(W, (int)e);
  } else {
    const int njo = d.Ko >> 4;
    const size_t total = (size_t)d.Kp * d.Ko;
    for (size_t e = e0; e < total; e += stride) {
      const int t = (int)(e & 3), col = (int)((e >> 2) & 15), kk = (int)((e >> 6) & 3);
      const size_t rest = e >> 8;
      const int jo = (int)(rest % njo), nt = (int)(rest / njo);
      const int o = 16 * jo + 4 * kk + t, k = 16 * nt + col;
      out[e] = (o < d.Cout && k < d.Cin) ? W[(size_t)o * d.Cin + k] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------ head: relu -> avgpool(k) -> (B, C, h, w) flatten
__global__ __launch_bounds__(256) void head_pool_kernel(const float* __restrict__ F, int ldf, int C, int B, int H, int W,
                                                        int k, float* __restrict__ out) {
  const int Ho = H / k, Wo = W / k;
  const size_t total = (size_t)B * Ho * Wo * C;
  const float inv = 1.0f / (float)(k * k);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t rest = e / C;
    const int ox = (int)(rest % Wo);
    rest /= Wo;
    const int oy = (int)(rest % Ho);
    const int b = (int)(rest / Ho);
    float s = 0.f;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j)
        s += fmaxf(F[((size_t)(b * H + oy * k + i) * W + ox * k + j) * ldf + c], 0.f);
    out[(((size_t)b * C + c) * Ho + oy) * Wo + ox] = s * inv;
  }
}

// ------------------------------------------------------------------------------ transition: pooled activation
// A[p'][c] = mean over the 2x2 window of relu(scale[c]*x + shift[c])  (avg-pool commutes with the 1x1 conv,
// DenseNet.py:14-21).  The transition's conv (3-4 output chunks) and its weight gradient then read A -- a quarter
// of the block buffer -- instead of re-reading and re-pooling the whole block buffer once per chunk.
__global__ __launch_bounds__(256) void pool_act_kernel(const float* __restrict__ X, int ldx, int B, int Hin, int Win,
                                                       int Kp, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ A, int lda,
                                                       unsigned short* __restrict__ relu_mask16) {
  const int Ho = Hin >> 1, Wo = Win >> 1, nq = Kp >> 2;
  const size_t total = (size_t)B * Ho * Wo * nq;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t pp = e / nq;
    const int q = (int)(e - pp * nq);
    const int b = (int)(pp / ((size_t)Ho * Wo)), rem = (int)(pp - (size_t)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const float* src = X + ((size_t)(b * Hin + 2 * oy) * Win + 2 * ox) * ldx + 4 * q;
    const float4 s4 = *reinterpret_cast<const float4*>(scale + 4 * q);
    const float4 t4 = *reinterpret_cast<const float4*>(shift + 4 * q);
    const float4 v0 = bn_relu4(*reinterpret_cast<const float4*>(src), s4, t4);
    const float4 v1 = bn_relu4(*reinterpret_cast<const float4*>(src + ldx), s4, t4);
    const float4 v2 = bn_relu4(*reinterpret_cast<const float4*>(src + (size_t)Win * ldx), s4, t4);
    const float4 v3 = bn_relu4(*reinterpret_cast<const float4*>(src + (size_t)Win * ldx + ldx), s4, t4);
    float4 a;  // same association as the fused-pool operand load of conv1x1_fwd_kernel<true>
    a.x = ((v0.x + v1.x) + (v2.x + v3.x)) * 0.25f;
    a.y = ((v0.y + v1.y) + (v2.y + v3.y)) * 0.25f;
    a.z = ((v0.z + v1.z) + (v2.z + v3.z)) * 0.25f;
    a.w = ((v0.w + v1.w) + (v2.w + v3.w)) * 0.25f;
    *reinterpret_cast<float4*>(A + pp * lda + 4 * q) = a;
    if (relu_mask16) {   // bit 4*sub + g <-> window pixel sub (row-major 2x2), channel 4q + g: the transition dgrad's ReLU mask
      auto nib = [](const float4& v) {
        return (unsigned)(v.x > 0.f) | ((unsigned)(v.y > 0.f) << 1) | ((unsigned)(v.z > 0.f) << 2) | ((unsigned)(v.w > 0.f) << 3);
      };
      relu_mask16[e] = (unsigned short)(nib(v0) | (nib(v1) << 4) | (nib(v2) << 8) | (nib(v3) << 12));
    }
  }
}

}  // namespace

// =============================================================================== C ABI
extern "C" int eml_dense_conv0_fwd_f32(const float* x, const float* w0, float* X, int ldx, int B, int H, int W,
                                       int C0, double* partials, int grid, eml_stream_t stream) {
  if (!x || !w0 || !X || !partials || B < 1 || H < 1 || W < 1 || grid < 1 || ldx < C0 || (ldx & 3))
    return eml::fail(EML_EINVAL, "eml_dense_conv0_fwd_f32: bad arguments");
  if (C0 != 24) return eml::fail(EML_EINVAL, "eml_dense_conv0_fwd_f32: only num_init_features=24 is built (got %d)", C0);
  hipLaunchKernelGGL(conv0_fwd_kernel<24>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w0, X, ldx, B, H, W, partials);
  return eml::check_launch("eml_dense_conv0_fwd_f32");
}

extern "C" int eml_dense_conv0_fwd_mfma_f32(const float* x, const float* w0, float* X, int ldx, int B, int H, int W,
                                            int C0, double* partials, int grid, eml_stream_t stream) {
  if (!x || !w0 || !X || !partials || B < 1 || H < 1 || W < 1 || grid < 1 || ldx < C0 || (ldx & 3) ||
      ((uintptr_t)X & 15))
    return eml::fail(EML_EINVAL, "eml_dense_conv0_fwd_mfma_f32: bad arguments");
  if (C0 != 24) return eml::fail(EML_EINVAL, "eml_dense_conv0_fwd_mfma_f32: only num_init_features=24 is built (got %d)", C0);
  if ((long)B * H * W * 3 > 2147483647L) return eml::fail(EML_EINVAL, "eml_dense_conv0_fwd_mfma_f32: input too large");
  hipLaunchKernelGGL(conv0_fwd_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w0, X, ldx, B, H, W, partials);
  return eml::check_launch("eml_dense_conv0_fwd_mfma_f32");
}

extern "C" int eml_dense_bn_apply_f32(const float* src, int ld_src, float* dst, int ld_dst, int C, long P,
                                      const float* scale, const float* shift, int relu, double* partials, int grid,
                                      eml_stream_t stream) {
  if (!src || !dst || !scale || !shift || !partials || C < 1 || C > 384 || P < 1 || grid < 1)
    return eml::fail(EML_EINVAL, "eml_dense_bn_apply_f32: bad arguments (C<=384)");
  if (C <= 32 && (C & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst) |
                                                                             reinterpret_cast<size_t>(scale) | reinterpret_cast<size_t>(shift)) & 15) == 0) {
    hipLaunchKernelGGL(bn_apply_small_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, C, (size_t)P,
                       scale, shift, relu, partials);
    return eml::check_launch("eml_dense_bn_apply_f32");
  }
  const size_t lds = (size_t)4 * ((C + 63) / 64) * 64 * 2 * sizeof(double);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, src, ld_src, dst, ld_dst, C,
                     (size_t)P, scale, shift, relu, partials);
  return eml::check_launch("eml_dense_bn_apply_f32");
}

extern "C" int eml_dense_bn_prepare_f32(const double* partials, int G, int pstride, int n_new, int c_new0, double count,
                                        float* mean, float* var, float* istd, const float* gamma, const float* beta,
                                        float* rmean, float* rvar, int C, int Cpad, float eps, float momentum,
                                        int training, float* scale, float* shift, eml_stream_t stream) {
  if (!mean || !var || !istd || count < 1.0) return eml::fail(EML_EINVAL, "eml_dense_bn_prepare_f32: bad arguments");
  if (scale && (!shift || !gamma || !beta || !rmean || !rvar || C < 1 || Cpad < C))
    return eml::fail(EML_EINVAL, "eml_dense_bn_prepare_f32: bad BN arguments");
  const int total = scale ? Cpad : n_new;
  if (total < 1) return EML_OK;
  hipLaunchKernelGGL(bn_prepare_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials, G, pstride, n_new, c_new0,
                     count, mean, var, istd, gamma, beta, rmean, rvar, C, Cpad, eps, momentum, training, scale, shift);
  return eml::check_launch("eml_dense_bn_prepare_f32");
}

extern "C" int eml_dense_permute_w1_f32(const float* W, int Cout, int Cin, int Kp, float* Wp, eml_stream_t stream) {
  if (!W || !Wp || Cout < 1 || Cin < 1 || Kp < Cin || (Kp & 15))
    return eml::fail(EML_EINVAL, "eml_dense_permute_w1_f32: bad arguments");
  const int nchunks = (Cout + 47) / 48;
  hipLaunchKernelGGL(permute_w1_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, W, Cout, Cin, Kp, nchunks, Wp);
  return eml::check_launch("eml_dense_permute_w1_f32");
}

extern "C" int eml_dense_permute_batch_f32(const eml_permute_desc* descs, int n, eml_stream_t stream) {
  if (!descs || n < 0 || n > 65535) return eml::fail(EML_EINVAL, "eml_dense_permute_batch_f32: bad arguments");
  if (n == 0) return EML_OK;
  hipLaunchKernelGGL(permute_batch_kernel, dim3(8, n), dim3(256), 0, (hipStream_t)stream, descs);
  return eml::check_launch("eml_dense_permute_batch_f32");
}

extern "C" int eml_dense_permute_w2_f32(const float* W2, int Cout, float* W2p, eml_stream_t stream) {
  if (!W2 || !W2p || Cout < 1 || Cout > 16) return eml::fail(EML_EINVAL, "eml_dense_permute_w2_f32: bad arguments");
  hipLaunchKernelGGL(permute_w2_kernel, dim3(27), dim3(256), 0, (hipStream_t)stream, W2, Cout, W2p);
  return eml::check_launch("eml_dense_permute_w2_f32");
}

// conv1x1 over all ceil(Cout/48) output chunks.  pool != 0: transition (2x2 avg-pool fused in
// the operand load; P = B*(Hin/2)*(Win/2) output pixels).  partials: [chunks][grid][48][2] f64.
extern "C" int eml_dense_conv1x1_fwd_f32(const float* X, int ldx, long P, int Hin, int Win, int pool, int Kp,
                                         const float* scale, const float* shift, const float* Wp, int Cout,
                                         float* out, int ldo, double* partials, int grid,
                                         unsigned long long* relu_mask, eml_stream_t stream) {
  if (!X || !scale || !shift || !Wp || !out || !partials || P < 1 || grid < 1 || Kp < 16 || (Kp & 15) || Kp > ldx ||
      (ldx & 3) || Cout < 1)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_fwd_f32: bad arguments");
  if (relu_mask && (pool || Cout > 48))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_fwd_f32: relu_mask is for dense layers (no pool, Cout <= 48)");
  if (pool && ((Hin & 1) || (Win & 1))) return eml::fail(EML_EINVAL, "eml_dense_conv1x1_fwd_f32: pool needs even H, W");
  // dense layers with a compact 48-wide output leave through a staging tile (whole-line stores): EML_FWD_STAGED=0 for the A/B
  static const bool staged_env = [] { const char* v = getenv("EML_FWD_STAGED"); return !(v && v[0] == '0'); }();
  const size_t lds0 = ((size_t)Kp * 48 + 2 * Kp) * sizeof(float) + 4 * 48 * 2 * sizeof(double) +
                      (relu_mask ? (size_t)4 * 16 * (Kp / 16) * sizeof(unsigned long long) : 0);
  const size_t stg_bytes = (size_t)4 * 16 * 52 * sizeof(float);
  // (only where two workgroups per CU still fit with the tile: the widest layers of blocks 2 and 3 keep the direct stores)
  const int staged = (staged_env && !pool && Cout == 48 && ldo == 48 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
                      lds0 + stg_bytes <= 80 * 1024) ? 1 : 0;
  const size_t lds = lds0 + (staged ? stg_bytes : 0);
  if (lds > 160 * 1024) return eml::fail(EML_EINVAL, "eml_dense_conv1x1_fwd_f32: Kp=%d does not fit LDS", Kp);
  const int nchunks = (Cout + 47) / 48;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int nv = (Cout - ch * 48 < 48) ? Cout - ch * 48 : 48;
    const float* wp = Wp + (size_t)ch * Kp * 48;
    double* pp = partials + (size_t)ch * grid * 96;
    if (pool) {
      EML_ENSURE_LDS((&conv1x1_fwd_kernel<true, false>), lds);
      hipLaunchKernelGGL((conv1x1_fwd_kernel<true, false>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X, ldx, (int)P, Hin,
                         Win, Kp, scale, shift, wp, out + ch * 48, ldo, nv, pp, nullptr, 0);
    } else if (relu_mask) {
      EML_ENSURE_LDS((&conv1x1_fwd_kernel<false, true>), lds);
      hipLaunchKernelGGL((conv1x1_fwd_kernel<false, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X, ldx, (int)P,
                         Hin, Win, Kp, scale, shift, wp, out + ch * 48, ldo, nv, pp, relu_mask, staged);
    } else {
      EML_ENSURE_LDS((&conv1x1_fwd_kernel<false, false>), lds);
      hipLaunchKernelGGL((conv1x1_fwd_kernel<false, false>), dim3(grid), dim3(256), lds, (hipStream_t)stream, X, ldx, (int)P,
                         Hin, Win, Kp, scale, shift, wp, out + ch * 48, ldo, nv, pp, nullptr, staged);
    }
    int rc = eml::check_launch("eml_dense_conv1x1_fwd_f32");
    if (rc) return rc;
  }
  return EML_OK;
}

extern "C" int eml_dense_conv3x3_fwd_f32(const float* Z, const float* scale2, const float* shift2, const float* W2p,
                                         float* X, int ldx, int c_out0, int B, int H, int W, double* partials, int grid,
                                         eml_stream_t stream) {
  if (!Z || !scale2 || !shift2 || !W2p || !X || !partials || B < 1 || H < 1 || W < 1 || grid < 1 || c_out0 + 12 > ldx || (c_out0 & 1) || (ldx & 1))
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_fwd_f32: bad arguments");
  const size_t lds = (size_t)(2 * kHH * kHW * kPS + 96) * sizeof(float) + 8 * 16 * 2 * sizeof(double);
  EML_ENSURE_LDS((&conv3x3_fwd_kernel), lds);
  hipLaunchKernelGGL(conv3x3_fwd_kernel, dim3(grid), dim3(kC3Threads), lds, (hipStream_t)stream, Z, scale2, shift2, W2p, X, ldx,
                     c_out0, B, H, W, partials);
  return eml::check_launch("eml_dense_conv3x3_fwd_f32");
}

extern "C" int eml_dense_pool_act_f32(const float* X, int ldx, int B, int Hin, int Win, int Kp, const float* scale,
                                      const float* shift, float* A, int lda, unsigned short* relu_mask16,
                                      eml_stream_t stream) {
  if (!X || !scale || !shift || !A || B < 1 || Hin < 2 || Win < 2 || (Hin & 1) || (Win & 1) || Kp < 4 || (Kp & 3) ||
      Kp > ldx || Kp > lda || (ldx & 3) || (lda & 3))
    return eml::fail(EML_EINVAL, "eml_dense_pool_act_f32: bad arguments");
  const size_t total = (size_t)B * (Hin / 2) * (Win / 2) * (Kp / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pool_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx, B, Hin, Win, Kp, scale, shift,
                     A, lda, relu_mask16);
  return eml::check_launch("eml_dense_pool_act_f32");
}

extern "C" int eml_dense_head_pool_fwd_f32(const float* F, int ldf, int C, int B, int H, int W, int k, float* out,
                                           eml_stream_t stream) {
  if (!F || !out || C < 1 || B < 1 || k < 1 || H < k || W < k) return eml::fail(EML_EINVAL, "eml_dense_head_pool_fwd_f32: bad arguments");
  const size_t total = (size_t)B * (H / k) * (W / k) * C;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(head_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, F, ldf, C, B, H, W, k, out);
  return eml::check_launch("eml_dense_head_pool_fwd_f32");
}
