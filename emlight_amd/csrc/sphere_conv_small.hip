// SphereConv2D (and the ordinary 3x3 convolution written as a gather) for the few-channel INPUT layers of the GenProjector:
// SPADE's mlp_shared 3 -> 128 + ReLU on the guide map (normalization.py:92-96, 28 of them per generator pass) and VGG19's
// conv1_1 3 -> 64 + ReLU.  (The discriminator's 6 -> 64 first stage stays on the general path: it always needs its input
// gradient, and with the masked dY formed for that anyway the general weight gradient measured faster.)
//
// These layers are HBM-bound on their OUTPUT: K = 9 * Cin = 27, so per pixel they write O floats for O * 2K flops.
// On the general path they were im2col (a 9x operand) + library GEMM + a separate activation pass, and in the backward an
// activation-backward pass, a bias-gradient reduction and a split-K batched GEMM -- 3 and 5 traversals of the (pixels, O)
// tensor.  Here: one traversal each way.
//   forward : wave = 32 pixels x all O channels; the 27 interpolated taps of a pixel are built in registers straight from
//             the tap table (x is a few MB: L2 resident), W2 lives in registers (56 VGPRs), f32 MFMA 16x16x4 with the channels
//             on the rows so that a lane's 4 results are 4 consecutive channels of one pixel (16-byte stores);
//             bias + leaky ReLU in the epilogue.
//   wgrad   : dW2[o][k] = sum_m g'[m][o] * A[m][k] with g' = dY * act'(Y) formed in registers from 16-byte loads of dY and Y
//             (the pixel axis is the MFMA's K dimension, so WHICH pixel a lane feeds is free: lanes take 4 consecutive channels
//             of "their" pixel and the four components go to four interleaved row tiles).  The padding column k = 9*Cin of A
//             is set to 1: its column of the result IS the bias gradient.  Per-wave partials are summed through LDS per
//             workgroup, then by a deterministic second kernel.
#include <algorithm>

#include "eml_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// interpolated tap value A[m][k] (k = tap * CIN + c), grid_sampler's corner order; out-of-range corners (index -1) contribute
// nothing.  All loads are UNCONDITIONAL (clamped address, the value selected afterwards): under `if (id >= 0)` each corner
// became its own basic block -- load, wait, add -- and a wave's 14 taps x 4 corners ran as 56 serial L2 round trips.
template <int CIN>
__device__ __forceinline__ float tap_value(const float* __restrict__ xb, const int* __restrict__ idx,
                                           const float* __restrict__ wgt, int p, int k) {
  const int tap = k / CIN, c = k - tap * CIN;
  const int4 id = *reinterpret_cast<const int4*>(idx + ((size_t)p * 9 + tap) * 4);
  const float4 w = *reinterpret_cast<const float4*>(wgt + ((size_t)p * 9 + tap) * 4);
  const float x0 = xb[(size_t)max(id.x, 0) * CIN + c], x1 = xb[(size_t)max(id.y, 0) * CIN + c];
  const float x2 = xb[(size_t)max(id.z, 0) * CIN + c], x3 = xb[(size_t)max(id.w, 0) * CIN + c];
  float v = (id.x >= 0 ? x0 : 0.f) * w.x;
  v += (id.y >= 0 ? x1 : 0.f) * w.y;
  v += (id.z >= 0 ? x2 : 0.f) * w.z;
  v += (id.w >= 0 ? x3 : 0.f) * w.w;
  return v;
}

template <int CIN, int O>
__global__ __launch_bounds__(256, 2) void sphere_conv_small_fwd_kernel(const float* __restrict__ X, const int* __restrict__ idx,
                                                                    const float* __restrict__ wgt,
                                                                    const float* __restrict__ W2 /*[O][9*CIN]*/,
                                                                    const float* __restrict__ bias, float* __restrict__ Y,
                                                                    int M, int HW, int Po, float slope) {
  // Round 6.  16 pixels per step; the four lanes (r, g = 0..3) of a pixel share its 9 taps (lane g: taps g, g + 4, g + 8) and
  // load a corner's CIN channels TOGETHER; the interpolated values go through a wave-private LDS tile A_l[pixel][9 CIN (+1)] into
  // the MFMA operand layout.  Round 4 gave every lane its own (tap, channel) COLUMNS: 28 scattered 4-byte gathers + 14 table
  // loads per lane and tile -- the texture-address unit, not HBM and not the matrix unit, bound the kernel (254 us for a
  // 128 x 256 layer whose 537 MB of output stream out in 100); now 12 + 6.  And a step's operands are requested a whole step
  // ahead of its MFMAs (table entries two ahead; two operand buffers of native vectors), W2 sits in LDS in fragment order:
  // Wl[(s * MT + mt) * 64 + lane] = W2[16 mt + r][4 s + g].
  constexpr int K = 9 * CIN, KS = (K + 3) / 4, MT = O / 16, LDA = 4 * KS;
  static_assert(CIN == 3, "a corner is loaded as three floats");
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  __shared__ float Wl[KS * MT * 64];
  __shared__ float A_all[4 * 16 * LDA];
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  float* A_l = A_all + (threadIdx.x >> 6) * 16 * LDA;
  for (int e = threadIdx.x; e < KS * MT * 64; e += 256) {
    const int l = e & 63, rest = e >> 6, mt = rest % MT, s_ = rest / MT;
    const int k = 4 * s_ + (l >> 4);
    Wl[e] = k < K ? W2[(size_t)(16 * mt + (l & 15)) * K + k] : 0.f;
  }
  for (int e = threadIdx.x; e < 4 * 16 * LDA; e += 256) A_all[e] = 0.f;   // (the padding columns K .. LDA - 1 stay zero)
  __syncthreads();
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float4 bq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    bq[mt] = bias ? *reinterpret_cast<const float4*>(bias + 16 * mt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int ntiles = (M + 15) / 16;
  constexpr int NJ = 3;   // taps per lane: g, g + 4, g + 8 (the last exists for g == 0 only)
  struct Tables {
    i32x4 id[NJ];
    int xoff, trow, ok;
  };
  struct Operands {
    f32x3 xr[NJ][4];
    f32x4 we[NJ];
    int ok;
  };
  auto tap_of = [&](int j) { return min(g + 4 * j, 8); };   // clamped: lanes without a third tap re-read tap 8 (not written)
  auto load_tables = [&](int t, Tables& tb) {
    const int m = t * 16 + r;
    tb.ok = (t < ntiles && m < M) ? 1 : 0;
    const int mc = tb.ok ? m : 0;
    const int b = mc / Po, p = mc - b * Po;
    tb.xoff = b * HW * CIN;
    tb.trow = p * 9;
#pragma unroll
    for (int j = 0; j < NJ; ++j) tb.id[j] = *reinterpret_cast<const i32x4*>(idx + ((size_t)p * 9 + tap_of(j)) * 4);
  };
  auto issue = [&](const Tables& tb, Operands& o) {
    o.ok = tb.ok;
    const float* xb = X + tb.xoff;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      o.we[j] = *reinterpret_cast<const f32x4*>(wgt + ((size_t)(tb.trow + tap_of(j))) * 4);   // 0 where the corner is off the map
#pragma unroll
      for (int e = 0; e < 4; ++e) o.xr[j][e] = *reinterpret_cast<const f32x3*>(xb + (size_t)max(tb.id[j][e], 0) * CIN);
    }
  };
  auto run = [&](int t, const Operands& o) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int tap = g + 4 * j;
      f32x3 v = o.xr[j][0] * o.we[j][0];     // grid_sampler's corner order, as tap_value
      v += o.xr[j][1] * o.we[j][1];
      v += o.xr[j][2] * o.we[j][2];
      v += o.xr[j][3] * o.we[j][3];
      if (!o.ok) v = f32x3{0.f, 0.f, 0.f};
      if (tap < 9) {
        A_l[r * LDA + CIN * tap + 0] = v[0];
        A_l[r * LDA + CIN * tap + 1] = v[1];
        A_l[r * LDA + CIN * tap + 2] = v[2];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private tile: LDS runs a wave's accesses in order
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const float av = A_l[r * LDA + 4 * s_ + g];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(Wl[(s_ * MT + mt) * 64 + lane], av, acc[mt]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int m = t * 16 + r;
    if (o.ok) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float4 v = make_float4(acc[mt][0] + bq[mt].x, acc[mt][1] + bq[mt].y, acc[mt][2] + bq[mt].z, acc[mt][3] + bq[mt].w);
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        *reinterpret_cast<float4*>(Y + (size_t)m * O + 16 * mt + 4 * g) = v;
      }
    }
  };
  Tables tab;
  Operands oa, ob;
  load_tables(wave, tab);
  issue(tab, oa);
  load_tables(wave + nwaves, tab);
  for (int t = wave; t < ntiles; t += 2 * nwaves) {
    issue(tab, ob);
    load_tables(t + 2 * nwaves, tab);
    __builtin_amdgcn_sched_barrier(0);
    run(t, oa);
    __builtin_amdgcn_sched_barrier(0);
    issue(tab, oa);
    load_tables(t + 3 * nwaves, tab);
    __builtin_amdgcn_sched_barrier(0);
    run(t + nwaves, ob);
    __builtin_amdgcn_sched_barrier(0);
  }
}


// Input gradient, first half: dA9[m][k] = sum_o g'[m][o] * W2[o][k]  (k = tap * CIN + c < 9 * CIN; g' = dY * act'(Y) formed in
// registers) -- the (M, 9 CIN) tensor eml_sphere_col2im_f32 then gathers into dX.  Needed where the 3-channel INPUT itself
// carries a gradient: the guide map of the joint step (18 mlp_shared convolutions per generator pass; emlight_amd/joint.py)
// and VGG19's conv1_1 on the generated panorama.  On the general path this was an activation-backward pass (read dY, Y; write
// g'), a library GEMM with N = 27 (read g' again) and, for the weights, a bias reduction, an im2col and a batched GEMM: here
// one read of (dY, Y).  MFMA: rows = k (two 16-row tiles), columns = 16 pixels, K = the output channels; a lane loads
// 4 consecutive channels of its pixel (16 bytes) and feeds them to four k-steps (MFMA's k index is only a summation label),
// W2 sits in registers as the A operand of those steps.
template <int CIN, int O>
__global__ __launch_bounds__(256, 2) void sphere_conv_small_da9_kernel(const float* __restrict__ dY, const float* __restrict__ Yact,
                                                                    const float* __restrict__ W2 /*[O][9*CIN]*/,
                                                                    float* __restrict__ dA9 /*[M][9*CIN]*/, int M, float slope) {
  constexpr int K = 9 * CIN, NTK = (K + 15) / 16, NS = O / 16;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float wreg[NTK][NS][4];
#pragma unroll
  for (int nt = 0; nt < NTK; ++nt)
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * nt + r;
        wreg[nt][s][i] = k < K ? W2[(size_t)(16 * s + 4 * g + i) * K + k] : 0.f;
      }
  const int ntiles = (M + 15) / 16;
  for (int t = wave; t < ntiles; t += nwaves) {
    const int m = t * 16 + r;
    const bool ok = m < M;
    const size_t row = (size_t)(ok ? m : 0) * O + 4 * g;
    float4 q[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) q[s] = *reinterpret_cast<const float4*>(dY + row + 16 * s);
    if (Yact) {   // wave-uniform
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float4 yq = *reinterpret_cast<const float4*>(Yact + row + 16 * s);
        q[s].x = yq.x > 0.f ? q[s].x : q[s].x * slope; q[s].y = yq.y > 0.f ? q[s].y : q[s].y * slope;
        q[s].z = yq.z > 0.f ? q[s].z : q[s].z * slope; q[s].w = yq.w > 0.f ? q[s].w : q[s].w * slope;
      }
    }
    f32x4 acc[NTK];
#pragma unroll
    for (int nt = 0; nt < NTK; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int nt = 0; nt < NTK; ++nt) {
        acc[nt] = mfma16(wreg[nt][s][0], q[s].x, acc[nt]);
        acc[nt] = mfma16(wreg[nt][s][1], q[s].y, acc[nt]);
        acc[nt] = mfma16(wreg[nt][s][2], q[s].z, acc[nt]);
        acc[nt] = mfma16(wreg[nt][s][3], q[s].w, acc[nt]);
      }
    if (ok) {   // D element i of tile nt: row 4g + i <-> k = 16 nt + 4g + i, column r <-> this lane's pixel
      float* dst = dA9 + (size_t)m * K;
#pragma unroll
      for (int nt = 0; nt < NTK; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = 16 * nt + 4 * g + i;
          if (k < K) dst[k] = acc[nt][i];
        }
    }
  }
}

// partial[blockIdx.x][O][KP]  (KP = 9*CIN + 1 rounded up to 16 columns: k < 9*CIN = dW2, k = 9*CIN = the bias gradient)
template <int CIN, int O>
__global__ __launch_bounds__(256, 2) void sphere_conv_small_wgrad_kernel(const float* __restrict__ X, const int* __restrict__ idx,
                                                                      const float* __restrict__ wgt,
                                                                      const float* __restrict__ dY, const float* __restrict__ Yact,
                                                                      float* __restrict__ partial, int M, int HW, int Po,
                                                                      float slope) {
  constexpr int K = 9 * CIN, NT = (K + 1 + 15) / 16, KP = 16 * NT, NG = O / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3][O * KP] wave partials of waves 1..3
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
  f32x4 acc[NG][4][NT];   // [64-channel group][component t of the float4][column tile]: row r <-> channel 64G + 4r + t
#pragma unroll
  for (int G = 0; G < NG; ++G)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[G][t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 4 pixels per MFMA k-step, U k-steps per step, and a step's operands are requested a whole step AHEAD of its MFMAs
  // (round 6).  A step is a two-level dependent chain -- the tap-table entry, then the four corners it names -- of ~2 us, in front
  // of 32 MFMAs: run step after step (round 4's loop, U sets of gathers in flight) every wave spent its time waiting for one
  // chain after the other: 331 us for the 1 M pixels of a 128 x 256 layer, and 67 us for the 1 K pixels of a 4 x 8 one (one
  // workgroup: see small_wgrad_grid).  Now the table entries of step i + 2 and the corners / dY / Y of step i + 1 are in
  // flight while step i's MFMAs run: two operand buffers (native vectors: arrays of HIP float4 structs that live across a
  // rolled loop are parked in scratch), the loop body written out for both, every load unconditional from a clamped pixel.
  constexpr int U = 2;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  struct Tables {   // entries of one step: (u, nt) -> this lane's pixel (st * U + u) * 4 + g, column 16 nt + r
    i32x4 id[U][NT];
    int xoff[U];    // sample offset into X (elements)
    int trow[U];    // row of the pixel in the tap table: the weights need no index and travel with the operands
    int okb;        // bit u: the pixel exists
  };
  struct Operands {
    f32x4 xr[U][NT];   // the four corners' values
    f32x4 we[U][NT];   // their weights (0 where the corner is off the map)
    f32x4 gq[U][NG], yq[U][NG];
    int okb;
  };
  const int nsteps = (M + 4 * U - 1) / (4 * U);
  int ccol[NT];     // this lane's column of tile nt -> channel of its tap (clamped column: padding lanes gather a valid address)
  int ctap[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int k = min(16 * nt + r, K - 1);
    ctap[nt] = k / CIN;
    ccol[nt] = k - ctap[nt] * CIN;
  }
  auto load_tables = [&](int st, Tables& t) {
    t.okb = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = (st * U + u) * 4 + g;
      const bool ok = st < nsteps && m < M;
      const int mc = ok ? m : 0;
      const int b = mc / Po, pp = mc - b * Po;
      t.xoff[u] = b * HW * CIN;
      t.trow[u] = pp * 9;
      t.okb |= ok ? (1 << u) : 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        t.id[u][nt] = *reinterpret_cast<const i32x4*>(idx + ((size_t)pp * 9 + ctap[nt]) * 4);
    }
  };
  auto issue_operands = [&](int st, const Tables& t, Operands& o) {
    o.okb = t.okb;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* xb = X + t.xoff[u] + 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const i32x4 id = t.id[u][nt];
        // (a corner off the map has index -1 AND weight 0 in the table -- eml_sphere_tap_table_f32 writes both; the clamped
        // address reads a finite value that the zero weight removes)
        o.we[u][nt] = *reinterpret_cast<const f32x4*>(wgt + ((size_t)(t.trow[u] + ctap[nt])) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o.xr[u][nt][e] = xb[(size_t)max(id[e], 0) * CIN + ccol[nt]];
      }
      const int m = (st * U + u) * 4 + g;
      const size_t row = (size_t)(((t.okb >> u) & 1) ? m : 0) * O;
#pragma unroll
      for (int G = 0; G < NG; ++G) {
        o.gq[u][G] = *reinterpret_cast<const f32x4*>(dY + row + 64 * G + 4 * r);
        if (Yact) o.yq[u][G] = *reinterpret_cast<const f32x4*>(Yact + row + 64 * G + 4 * r);   // wave-uniform
      }
    }
  };
  auto run_step = [&](const Operands& o) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = (o.okb >> u) & 1;
      float bv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int k = 16 * nt + r;
        float v = o.xr[u][nt][0] * o.we[u][nt][0];     // grid_sampler's corner order, as tap_value
        v += o.xr[u][nt][1] * o.we[u][nt][1];
        v += o.xr[u][nt][2] * o.we[u][nt][2];
        v += o.xr[u][nt][3] * o.we[u][nt][3];
        bv[nt] = !ok ? 0.f : k < K ? v : k == K ? 1.f : 0.f;
      }
#pragma unroll
      for (int G = 0; G < NG; ++G) {
        f32x4 q = o.gq[u][G];
        if (Yact) {
          const f32x4 yq = o.yq[u][G];
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = yq[e] > 0.f ? q[e] : q[e] * slope;
        }
        if (!ok) q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[G][0][nt] = mfma16(q[0], bv[nt], acc[G][0][nt]);
          acc[G][1][nt] = mfma16(q[1], bv[nt], acc[G][1][nt]);
          acc[G][2][nt] = mfma16(q[2], bv[nt], acc[G][2][nt]);
          acc[G][3][nt] = mfma16(q[3], bv[nt], acc[G][3][nt]);
        }
      }
    }
  };
  Tables tab;
  Operands opA, opB;
  load_tables(wave, tab);
  issue_operands(wave, tab, opA);
  load_tables(wave + nwaves, tab);
  for (int st = wave; st < nsteps; st += 2 * nwaves) {
    // steps past the end have no pixel: their loads go to pixel 0 and their MFMAs add zeros (the tail of a wave's last turn)
    issue_operands(st + nwaves, tab, opB);
    load_tables(st + 2 * nwaves, tab);
    __builtin_amdgcn_sched_barrier(0);
    run_step(opA);
    __builtin_amdgcn_sched_barrier(0);
    issue_operands(st + 2 * nwaves, tab, opA);
    load_tables(st + 3 * nwaves, tab);
    __builtin_amdgcn_sched_barrier(0);
    run_step(opB);
    __builtin_amdgcn_sched_barrier(0);
  }
  // D element i of tile (G, t, nt): row 4g + i <-> channel 64G + 4(4g + i) + t, column 16nt + r
  if (wv > 0) {
    float* dst = smem + (size_t)(wv - 1) * O * KP;
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[(64 * G + 4 * (4 * g + i) + t) * KP + 16 * nt + r] = acc[G][t][nt][i];
  }
  __syncthreads();
  if (wv == 0) {
    float* out = partial + (size_t)blockIdx.x * O * KP;
#pragma unroll
    for (int G = 0; G < NG; ++G)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = (64 * G + 4 * (4 * g + i) + t) * KP + 16 * nt + r;
            out[e] = ((acc[G][t][nt][i] + smem[e]) + smem[O * KP + e]) + smem[2 * O * KP + e];
          }
  }
}

// dW2[o][k] (k < K) and db[o] (column K) = fixed-order sum of the workgroup partials.  Block = 16 elements x 16 partial lanes
// (a 16-thread-wide reduction per element would leave 16 workgroups summing 512 partials each: 36 us)
__global__ __launch_bounds__(256) void small_wgrad_reduce_kernel(const float* __restrict__ partial, int S, int O, int K, int KP,
                                                                 float* __restrict__ dW2, float* __restrict__ db) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < O * KP)
    for (int z = zl; z < S; z += 16) s += partial[(size_t)z * O * KP + e];
  red[zl][el] = s;
  __syncthreads();
  if (zl == 0 && e < O * KP) {
    float tot = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) tot += red[z][el];
    const int o = e / KP, k = e - o * KP;
    if (k < K) dW2[(size_t)o * K + k] = tot;
    else if (k == K && db) db[o] = tot;
  }
}

constexpr int kSmallGrid = 512;   // persistent workgroups of the weight gradient (= its partial count)

template <int CIN, int O>
void launch_small_wgrad(const float* X, const int* idx, const float* wgt, const float* dY, const float* Yact, float* partial,
                        float* dW2, float* db, int M, int HW, int Po, float slope, int grid, hipStream_t st) {
  constexpr int K = 9 * CIN, NT = (K + 1 + 15) / 16, KP = 16 * NT;
  const size_t lds = (size_t)3 * O * KP * sizeof(float);
  EML_ENSURE_LDS((&sphere_conv_small_wgrad_kernel<CIN, O>), lds);
  hipLaunchKernelGGL((sphere_conv_small_wgrad_kernel<CIN, O>), dim3(grid), dim3(256), lds, st, X, idx, wgt, dY, Yact, partial, M,
                     HW, Po, slope);
  hipLaunchKernelGGL(small_wgrad_reduce_kernel, dim3((O * KP + 15) / 16), dim3(256), 0, st, partial, grid, O, K, KP, dW2, db);
}

int small_kp(int C) { return 16 * ((9 * C + 1 + 15) / 16); }
bool small_supported(int C, int O) { return C == 3 && (O == 64 || O == 128); }
// one workgroup per 128 pixels up to the persistent grid (round 6; it was one per 1024: the 1 K pixels of a 4 x 8 layer ran on
// ONE workgroup, 32 dependent steps per wave = 67 us of latency for 26 MFLOP)
int small_wgrad_grid(long M) { return (int)std::min<long>(kSmallGrid, std::max<long>(1, (M + 127) / 128)); }

}  // namespace

extern "C" int eml_sphere_conv_small_supported(int C, int O) { return small_supported(C, O) ? 1 : 0; }

extern "C" int eml_sphere_conv_small_fwd_f32(const float* X, const int* idx, const float* wgt, const float* W2,
                                             const float* bias, float* Y, int B, int HW, int Po, int C, int O,
                                             float act_slope, eml_stream_t stream) {
  if (!X || !idx || !wgt || !W2 || !Y || B < 0 || HW < 1 || Po < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_fwd_f32: null pointer or empty shape");
  if (!small_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_fwd_f32: (C, O) = (%d, %d) not in {(3, 64), (3, 128)}", C, O);
  if (!(act_slope >= 0.f && act_slope <= 1.f))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_fwd_f32: act_slope %g outside [0, 1]", (double)act_slope);
  const long M = (long)B * Po;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_small_fwd_f32: too many pixels");
  if (M == 0) return EML_OK;
  const int grid = (int)std::min<long>(1024, (M + 127) / 128);   // persistent: W2 is loaded into registers once per wave
  hipStream_t st = (hipStream_t)stream;
#define EML_SMALL_FWD(CV, OV)                                                                                              \
  hipLaunchKernelGGL((sphere_conv_small_fwd_kernel<CV, OV>), dim3(grid), dim3(256), 0, st, X, idx, wgt, W2, bias, Y, (int)M, HW, \
                     Po, act_slope)
  if (O == 128) EML_SMALL_FWD(3, 128);
  else EML_SMALL_FWD(3, 64);
#undef EML_SMALL_FWD
  return eml::check_launch("eml_sphere_conv_small_fwd_f32");
}

extern "C" size_t eml_sphere_conv_small_wgrad_partial_floats(int B, int Po, int C, int O) {
  if (!small_supported(C, O) || B < 1 || Po < 1) return 0;
  return (size_t)small_wgrad_grid((long)B * Po) * O * small_kp(C);
}

extern "C" int eml_sphere_conv_small_wgrad_f32(const float* X, const int* idx, const float* wgt, const float* dY,
                                               const float* Yact, float act_slope, float* partial, float* dW2, float* db,
                                               int B, int HW, int Po, int C, int O, eml_stream_t stream) {
  if (!X || !idx || !wgt || !dY || !partial || !dW2 || B < 0 || HW < 1 || Po < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_wgrad_f32: null pointer or empty shape");
  if (!small_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_wgrad_f32: (C, O) = (%d, %d) not in {(3, 64), (3, 128)}", C, O);
  if (!(act_slope >= 0.f && act_slope <= 1.f) || (act_slope != 1.f && !Yact))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_wgrad_f32: act_slope %g needs Yact and a slope in [0, 1]", (double)act_slope);
  const long M = (long)B * Po;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_small_wgrad_f32: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    (void)hipMemsetAsync(dW2, 0, (size_t)O * 9 * C * sizeof(float), st);
    if (db) (void)hipMemsetAsync(db, 0, (size_t)O * sizeof(float), st);
    return eml::check_launch("eml_sphere_conv_small_wgrad_f32");
  }
  const int grid = small_wgrad_grid(M);
  const float* ya = act_slope != 1.f ? Yact : nullptr;
  if (O == 128) launch_small_wgrad<3, 128>(X, idx, wgt, dY, ya, partial, dW2, db, (int)M, HW, Po, act_slope, grid, st);
  else launch_small_wgrad<3, 64>(X, idx, wgt, dY, ya, partial, dW2, db, (int)M, HW, Po, act_slope, grid, st);
  return eml::check_launch("eml_sphere_conv_small_wgrad_f32");
}

extern "C" int eml_sphere_conv_small_da9_f32(const float* dY, const float* Yact, float act_slope, const float* W2, float* dA9,
                                             long M, int C, int O, eml_stream_t stream) {
  if (!dY || !W2 || !dA9 || M < 0) return eml::fail(EML_EINVAL, "eml_sphere_conv_small_da9_f32: null pointer or negative row count");
  if (!small_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_da9_f32: (C, O) = (%d, %d) not in {(3, 64), (3, 128)}", C, O);
  if (!(act_slope >= 0.f && act_slope <= 1.f) || (act_slope != 1.f && !Yact))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_small_da9_f32: act_slope %g needs Yact and a slope in [0, 1]", (double)act_slope);
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_small_da9_f32: too many pixels");
  if (M == 0) return EML_OK;
  const int grid = (int)std::min<long>(2048, (M + 63) / 64);   // persistent: W2 is loaded into registers once per wave
  const float* ya = act_slope != 1.f ? Yact : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (O == 128)
    hipLaunchKernelGGL((sphere_conv_small_da9_kernel<3, 128>), dim3(grid), dim3(256), 0, st, dY, ya, W2, dA9, (int)M, act_slope);
  else
    hipLaunchKernelGGL((sphere_conv_small_da9_kernel<3, 64>), dim3(grid), dim3(256), 0, st, dY, ya, W2, dA9, (int)M, act_slope);
  return eml::check_launch("eml_sphere_conv_small_da9_f32");
}
