// Backward kernels of the DenseNet-BC encoder for gfx950 (f32 MFMA 16x16x4).
//
// Per dense layer (reverse order), with G the gradient buffer that mirrors the block buffer X:
//   conv3x3_bwd_data   dzn = conv2^T(g), g = G[:, Cin:Cin+12] + sB*x + sC (deferred BN1 affine applied while
//                      staging; g also written compactly for the weight gradient) + partial (sum dzn, sum dzn*zhat)
//   conv3x3_bwd_weight dW2 = sum_p g[p] (x) BN2(z)[p+tap]        (persistent accumulators, partials)
//   bn_bwd_finalize    dgamma2, dbeta2, and the affine that turns dzn into dz on the fly:
//                      dz = cA*dzn + cB*z + cC   (BatchNorm backward is affine per channel)
//   conv1x1_bwd_weight dW1 = sum_p relu(bn1(x))[p] (x) dz[p]     (dz rebuilt in the operand load and MATERIALISED
//                      in place over dzn: the data-gradient passes read 48 floats per pixel, not dzn + z = 96)
//   conv1x1_bwd_narrow the upper layer of a pair: its data gradient over the lower layer's 12 output channels, added
//                      to the G slice and written as a compact (P,12) tensor that the lower layer's conv3x3_bwd_data
//                      stages instead of the slice
//   conv1x1_bwd_data_multi   dam = relu-mask * (dz W1) for ONE or TWO consecutive layers per pass over X;
//                      G[:, :Cin] += scale1*dam in the epilogue (the dy-coefficient of BN1's backward needs no
//                      statistics) + partial (sum, sum*xhat) per layer
//   bn_bwd_finalize    dgamma1, dbeta1; the statistics-dependent affine (cB*x + cC) is only summed
//                      per channel into sB/sC and applied ONCE per channel, where the channel is consumed
//                      (conv3x3_bwd_data; grad_materialize for a block's input channels)
// The transitions: weight gradient on the pooled activation the forward kept (pool_act), data gradient by
// transition_bwd_data (un-pool, mask, accumulate), last_norm's backward folded into the dz affine.  All
// reductions are per-block partials + a finishing kernel: deterministic, no atomics.
#include "eml_common.h"

#include <cstdint>
#include <cstdlib>
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// component t of a float4 (t is a compile-time constant after unrolling).  MFMA loops run t OUTERMOST so
// that consecutive MFMAs hit different accumulators: back-to-back MFMAs on one accumulator pay the
// 40-cycle dependent latency instead of the 32-cycle issue interval (MI355X_MICROARCH.md).
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

constexpr int kTH = 8, kTW = 32, kHH = kTH + 2, kHW = kTW + 2;

// =============================================================================== conv3x3 backward: data
// dzn[q][c] = sum_{dy,dx,o} W2[o][c][dy][dx] * g[(qy-dy+1, qx-dx+1)][o]
// 512 threads = 8 waves, one output row of the 8x32 tile per wave (2 pixel tiles x 3 channel tiles of
// accumulators), one workgroup per CU with 2 waves per SIMD.  Everything a tile needs from HBM is requested
// inside the MFMA stream of the tile before it: the next g halo tile (registers -> second LDS buffer) and this
// tile's z rows for the BN2 statistics, so a tile costs one barrier and no exposed HBM round trip.  The kernel
// is HBM-bound (g 48 B + z 192 B in, dzn 192 B out per pixel against 81 MFMAs per 16 pixels).
constexpr int kPSG = 14;  // LDS pixel stride of the g halo tile: 14 dwords makes the ds_read_b32 lane groups conflict-free
constexpr int kBD = 512;
constexpr int kGRows = 2;                                   // halo rows staged per pass
constexpr int kGPass = kHH / kGRows;                        // 5 passes of float2 loads per thread

// FUSE: g = G + sB*X + sC (the deferred BN1 affine of the block gradient, see grad_materialize_kernel) is applied
// while the halo tile is staged, and the finished 12-channel gradient of the tile's own pixels is written to the
// compact GF (P,12) for conv3x3_bwd_weight -- instead of a separate read-modify-write pass over G.
// G may be the block gradient (ldg = ld, c0 = the layer's channel offset) or the compact (P,12) tensor that
// conv1x1_bwd_narrow_kernel leaves for the lower layer of a pair (ldg = 12, c0 = 0); cx is the layer's channel
// offset in X / sB / sC either way.
// WIDE (round 4): the halo tile is staged as 340 pixels x 3 float4 = 1020 items, two per thread, instead of five float2
// passes over (2 rows x 34 columns x 6 float2) -- 4 (FUSE: with x) 16-byte loads per thread and tile instead of 10 eight-
// byte ones, and the compact GF leaves as float4.  Needs the channel offsets to be multiples of 4 (blocks 1 and 2 of
// EMLight's encoder; block 3 starts at channel 150 and keeps the float2 path).
// TH (round 4): output rows per tile = waves per workgroup.  8 (512 threads, one workgroup per CU) is round 1's geometry and
// the one that runs.  4 (256 threads, TWO independent workgroups per CU, EML_D3_SHORT=1) was the experiment "with one
// workgroup per CU the two waves of a SIMD share a phase, so nobody issues MFMAs while both sit in the epilogue": measured
// 15.34 against 14.43 ms per step -- SLOWER.  The kernel moves 4.0 TB/s with 43 % of its bytes written (dzn, GF): it sits on
// what this part delivers for that read / write mix (4.0-4.7 TB/s, profiles/r03_row_access_probe.txt), not on its phases.
template <bool FUSE, bool WIDE, int TH>
__global__ __launch_bounds__(TH * 64, 2) void conv3x3_bwd_data_kernel(
    const float* __restrict__ G, int ldg, int c0, const float* __restrict__ W2, const float* __restrict__ Z,
    const float* __restrict__ zmean, const float* __restrict__ zistd, float* __restrict__ DZ, int B, int H, int W,
    double* __restrict__ partials /*[grid][48][2]*/, const float* __restrict__ Xb, int ldx, int cx,
    const float* __restrict__ sB, const float* __restrict__ sC, float* __restrict__ GF) {
  constexpr int HH = TH + 2, NT = TH * 64;
  constexpr int GR = TH == 8 ? 2 : 1;                       // halo rows staged per pass of the float2 path
  constexpr int kNarrowPass = HH / GR;
  __shared__ __attribute__((aligned(16))) float g_l[2][HH * kHW * kPSG];
  __shared__ double red[TH * 48 * 2];
  __shared__ __attribute__((aligned(16))) float coef_l[24];   // WIDE + FUSE: sB | sC of the layer's 12 channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  if constexpr (FUSE && WIDE) {
    if (tid < 24) coef_l[tid] = tid < 12 ? sB[cx + tid] : sC[cx + tid - 12];   // visible after the barrier below the prologue
  }

  // A fragments (D^T form): lane (kk, c = 16n + r) holds W2[o = 4s + kk][c][tap]
  float bw[9][3][3];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int n = 0; n < 3; ++n) bw[tap][s][n] = W2[((size_t)(4 * s + kk) * 48 + 16 * n + r) * 9 + tap];

  const int tx_n = (W + kTW - 1) / kTW, ty_n = (H + TH - 1) / TH;
  const int ntiles = B * ty_n * tx_n;
  double s1[3][4], s2[3][4];  // sum dzn, sum dzn*z  (xhat is applied to the f64 totals at the end)
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) s1[n][g] = s2[n][g] = 0.0;

  // staging map (narrow): GR halo rows x 34 columns x 6 float2 per pass (408 / 204 threads; the rest duplicate the last item)
  // staging map (WIDE): item t = tid + NT * it (< HH * 34 * 3, the tail duplicates the last): halo pixel t / 3, float4 t % 3
  constexpr int kWideItems = HH * kHW * 3;
  constexpr int kPass = WIDE ? (kWideItems + NT - 1) / NT : kNarrowPass;
  const int st = min(tid, GR * kHW * 6 - 1);
  const int s_row = st / (kHW * 6), s_rem = st - s_row * (kHW * 6);
  const int s_hx = s_rem / 6, s_q = s_rem - 6 * s_hx;
  const int s_dst = (s_row * kHW + s_hx) * kPSG + 2 * s_q;
  float2 gt[WIDE ? 1 : kPass], xt[(FUSE && !WIDE) ? kPass : 1];
  float4 gt4[WIDE ? kPass : 1], xt4[(FUSE && WIDE) ? kPass : 1];
  float2 fb = make_float2(0.f, 0.f), fc = make_float2(0.f, 0.f);
  constexpr int kWP = WIDE ? kPass : 1;
  int w_hy[kWP], w_hx[kWP], w_q[kWP];
  if constexpr (WIDE) {
#pragma unroll
    for (int it = 0; it < kWP; ++it) {
      const int t = min(tid + NT * it, kWideItems - 1), hp = t / 3;
      w_q[it] = t - 3 * hp;
      w_hy[it] = hp / kHW;
      w_hx[it] = hp - w_hy[it] * kHW;
    }
  } else if constexpr (FUSE) {
    fb = *reinterpret_cast<const float2*>(sB + cx + 2 * s_q);
    fc = *reinterpret_cast<const float2*>(sC + cx + 2 * s_q);
  }
  const float* s_src = G;
  const float* s_srcx = Xb;
  float* s_gf = GF;
  int s_y0 = 0;
  bool s_col = false, s_own = false;
  int w_pix[kWP];          // clamped source pixel of the item (addresses are formed at the load: registers are scarce here)
  bool w_ok[kWP], w_own[kWP];
  auto stage_begin = [&](int tile) {
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
    if constexpr (WIDE) {
#pragma unroll
      for (int it = 0; it < kWP; ++it) {
        const int gy = ty * TH - 1 + w_hy[it], gx = tx * kTW - 1 + w_hx[it];
        w_ok[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        w_pix[it] = (b * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1);
        if constexpr (FUSE) {
          // a pixel of the tile itself (not halo); the duplicated tail items of pass 1 write the same value twice
          w_own[it] = w_ok[it] && w_hy[it] >= 1 && w_hy[it] <= TH && w_hx[it] >= 1 && w_hx[it] <= kTW;
        }
      }
      return;
    }
    const int gx = tx * kTW - 1 + s_hx;
    s_y0 = ty * TH - 1 + s_row;
    s_col = gx >= 0 && gx < W;
    const size_t col = (size_t)b * H * W + min(max(gx, 0), W - 1);
    s_src = G + col * ldg + c0 + 2 * s_q;
    if constexpr (FUSE) {
      s_srcx = Xb + col * ldx + cx + 2 * s_q;
      s_gf = GF + col * 12 + 2 * s_q;
      s_own = s_col && tid < GR * kHW * 6 && s_hx >= 1 && s_hx <= kTW;  // a column of the tile itself (not halo)
    }
  };
  auto stage_load = [&](int it) {  // unconditional (clamped): exec-masked loads make the compiler stall MFMAs on them
    if constexpr (WIDE) {
      gt4[it] = *reinterpret_cast<const float4*>(G + (size_t)w_pix[it] * ldg + c0 + 4 * w_q[it]);
      if constexpr (FUSE) xt4[it] = *reinterpret_cast<const float4*>(Xb + (size_t)w_pix[it] * ldx + cx + 4 * w_q[it]);
      return;
    }
    const size_t row = (size_t)min(max(s_y0 + GR * it, 0), H - 1) * W;
    gt[it] = *reinterpret_cast<const float2*>(s_src + row * ldg);
    if constexpr (FUSE) xt[it] = *reinterpret_cast<const float2*>(s_srcx + row * ldx);
  };
  auto stage_commit = [&](int it, float* dst) {
    if constexpr (WIDE) {
      float4 v = gt4[it];
      if constexpr (FUSE) {
        const float4 fb4 = *reinterpret_cast<const float4*>(coef_l + 4 * w_q[it]);
        const float4 fc4 = *reinterpret_cast<const float4*>(coef_l + 12 + 4 * w_q[it]);
        v.x = fmaf(fb4.x, xt4[it].x, v.x) + fc4.x;
        v.y = fmaf(fb4.y, xt4[it].y, v.y) + fc4.y;
        v.z = fmaf(fb4.z, xt4[it].z, v.z) + fc4.z;
        v.w = fmaf(fb4.w, xt4[it].w, v.w) + fc4.w;
      }
      if (!w_ok[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      float* d = dst + (w_hy[it] * kHW + w_hx[it]) * kPSG + 4 * w_q[it];   // 56-byte pixel stride: 8-byte aligned
      *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
      if constexpr (FUSE) {
        if (w_own[it]) *reinterpret_cast<float4*>(GF + (size_t)w_pix[it] * 12 + 4 * w_q[it]) = v;
      }
      return;
    }
    const int gy = s_y0 + GR * it;
    const bool ok = s_col && gy >= 0 && gy < H;
    float2 v;
    if constexpr (FUSE) {
      v.x = ok ? fmaf(fb.x, xt[it].x, gt[it].x) + fc.x : 0.f;
      v.y = ok ? fmaf(fb.y, xt[it].y, gt[it].y) + fc.y : 0.f;
    } else {
      v.x = ok ? gt[it].x : 0.f;
      v.y = ok ? gt[it].y : 0.f;
    }
    *reinterpret_cast<float2*>(dst + s_dst + it * GR * kHW * kPSG) = v;
    if constexpr (FUSE) {
      const int hy = s_row + GR * it;  // halo row 0..HH-1; rows 1..TH are the tile's own
      if (s_own && ok && hy >= 1 && hy <= TH) *reinterpret_cast<float2*>(s_gf + (size_t)gy * W * 12) = v;
    }
  };

  int tile = blockIdx.x, cur = 0;
  if constexpr (FUSE && WIDE) __syncthreads();   // coef_l
  if (tile < ntiles) {
    stage_begin(tile);
#pragma unroll
    for (int it = 0; it < kPass; ++it) stage_load(it);
#pragma unroll
    for (int it = 0; it < kPass; ++it) stage_commit(it, g_l[0]);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): bw[] is complete on every path into the loop (see conv3x3_fwd)
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    stage_begin(nxt < ntiles ? nxt : tile);
    const float* gc = g_l[cur];
    float* gn = g_l[cur ^ 1];
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
    const int gy = ty * TH + wave;
    // this lane's two output pixels (row gy, columns 16m + r); clamped copies for the unconditional z loads
    size_t prow[2];
    bool pv[2];
    float4 zr[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int gx = tx * kTW + 16 * m + r;
      pv[m] = gy < H && gx < W;
      prow[m] = ((size_t)(b * H + min(gy, H - 1)) * W + min(gx, W - 1)) * 48;
    }

    f32x4 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float aq[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) aq[0][m] = gc[((wave + 2) * kHW + 16 * m + r + 2) * kPSG + kk];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int gi = tap * 3 + s;
        if (gi < kPass) {
          stage_load(gi);
          __builtin_amdgcn_sched_barrier(0);  // keep the request here; the scheduler would sink it to its use
        } else if (gi < kPass + 6) {
          const int zi = gi - kPass;
          zr[zi / 3][zi % 3] = *reinterpret_cast<const float4*>(Z + prow[zi / 3] + 16 * (zi % 3) + 4 * kk);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (gi + 1 < 27) {   // the g operand of group gi + 1 is requested before the MFMAs of group gi (round 4)
          const int tn = (gi + 1) / 3, sn = (gi + 1) - 3 * tn, dyn = tn / 3, dxn = tn - 3 * dyn;
#pragma unroll
          for (int m = 0; m < 2; ++m)
            aq[(gi + 1) & 1][m] = gc[((wave + 2 - dyn) * kHW + 16 * m + r + 2 - dxn) * kPSG + 4 * sn + kk];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[m][n] = mfma16(bw[tap][s][n], aq[gi & 1][m], acc[m][n]);  // D[channel][pixel]
        if (gi >= 27 - kPass) {
          __builtin_amdgcn_sched_barrier(0);
          stage_commit(gi - (27 - kPass), gn);
        }
      }
    }
    // epilogue: lane owns channels 16n + 4kk .. +3 of its two pixels.  Statistics first (they wait for the z
    // loads), stores after a scheduling fence: vmcnt counts in order, so a z wait placed after the stores
    // would wait for the stores too.  The z registers are consumed unconditionally (masked value): a load whose
    // only use is exec-masked stays "pending" for the compiler and costs a vmcnt(0) at the top of the next tile.
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      float l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float4 z = zr[m][n];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v = pv[m] ? acc[m][n][g] : 0.f;
          l1[g] += v;
          l2[g] = fmaf(v, f4c(z, g), l2[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s1[n][g] += (double)l1[g];
        s2[n][g] += (double)l2[g];
      }
    }
    // vmcnt(0) HERE (only the z loads are outstanding, and they were requested a whole MFMA stream ago): the
    // compiler sinks the statistics below the exec-masked stores, and its own wait there would cover the stores.
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int m = 0; m < 2; ++m)
      if (pv[m]) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
          *reinterpret_cast<float4*>(DZ + prow[m] + 16 * n + 4 * kk) =
              make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
      }
    eml::lds_barrier();  // buffer cur^1 is complete; everyone is done reading buffer cur (LDS only: no wait for the stores)
    cur ^= 1;
  }
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s1[n][g] += shfl_xor_d(s1[n][g], o);
        s2[n][g] += shfl_xor_d(s2[n][g], o);
      }
      if (r == 0) {
        red[(wave * 48 + 16 * n + 4 * kk + g) * 2 + 0] = s1[n][g];
        red[(wave * 48 + 16 * n + 4 * kk + g) * 2 + 1] = s2[n][g];
      }
    }
  __syncthreads();
  if (tid < 48) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < TH; ++w8) {
      t1 += red[(w8 * 48 + tid) * 2 + 0];
      t2 += red[(w8 * 48 + tid) * 2 + 1];
    }
    // sum dzn*zhat = istd * (sum dzn*z - mean * sum dzn), in f64
    partials[(size_t)blockIdx.x * 96 + 2 * tid + 0] = t1;
    partials[(size_t)blockIdx.x * 96 + 2 * tid + 1] = (double)zistd[tid] * (t2 - (double)zmean[tid] * t1);
  }
}

// =============================================================================== conv3x3 backward: weight
// dW2[o][c][dy][dx] = sum_p g[p][o] * zn[p + (dy-1, dx-1)][c];  D[i=c][j=o], MFMA-k = pixel.
// 512 threads = 8 waves: waves 0-3 contract the upper 4 rows of the 8x32 tile, waves 4-7 the lower 4; within a
// half the 27 (tap, 16-channel group) accumulator tiles are spread over the 4 waves (7,7,7,6).  Each half writes
// its own partial row.  The BN2(z) halo tile is double-buffered in LDS and, like the g operand registers, is
// refilled for the NEXT tile from inside this tile's MFMA stream (see conv3x3_fwd_kernel); g[ks] is reloaded in
// place right after its last use.
constexpr int kPSW = 48;   // 48 dwords: the two pixels of a ds_read_b32 lane group fall on disjoint bank halves
constexpr int kBW = 512;
constexpr int kGL = kTH * kTW * 12;   // floats of a tile's own g pixels in LDS ([pixel][12]: two pixels of a ds_read_b32 lane
                                      // group sit 12 banks apart -- disjoint for the 12 channels)

// GLDS (round 4): the g operand (k = pixel, lane (kk, o) needs g[pixel 4ks + kk][o] for 32 k-steps) comes from an LDS copy
// of the tile's 256 x 12 own pixels, staged by TWO 16-byte loads per thread, instead of 32 four-byte loads per lane and
// tile -- 42 vector-memory instructions per wave and tile become 12.  (Round 4's gather-GEMM measurements priced a
// vector-memory instruction at ~45 matrix-pipe cycles whatever it fetches: it holds the issuing wave ~60 cycles.)
// GLDS = false keeps the round-1 register path for the A/B (EML_W3_GREG=1).
template <bool GLDS>
__global__ __launch_bounds__(kBW) void conv3x3_bwd_weight_kernel(
    const float* __restrict__ G, int ldg, int c0, const float* __restrict__ Z, const float* __restrict__ scale2,
    const float* __restrict__ shift2, int B, int H, int W, float* __restrict__ partial /*[2*grid][27][16][16]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][kHH*kHW][48]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int half = wave >> 2, w4 = wave & 3;
  f32x4 acc[7];
  int aoff[7];  // LDS offset of this wave's (tap, mc) pair relative to the pixel, plus the lane's own pixel
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int idx = min(w4 + 4 * i, 26), tap = idx / 3, mc = idx - 3 * tap;
    aoff[i] = ((tap / 3 + 4 * half) * kHW + (tap % 3) + kk) * kPSW + 16 * mc + r;
  }
  const int tx_n = (W + kTW - 1) / kTW, ty_n = (H + kTH - 1) / kTH;
  const int ntiles = B * ty_n * tx_n;

  // z staging map (as conv3x3_fwd): thread = (halo column, 16-byte slice), one halo row per pass
  const int s_hx = min(tid / 12, kHW - 1), s_q = tid % 12;
  const float4 sc = *reinterpret_cast<const float4*>(scale2 + 4 * s_q);
  const float4 sh = *reinterpret_cast<const float4*>(shift2 + 4 * s_q);
  const int s_dst = s_hx * kPSW + 4 * s_q;
  float4 zt[kHH];
  const float* s_src = Z;
  int s_y0 = 0;
  bool s_col = false;
  // g operand of the tile being prefetched: lane (kk, o = r) needs g[pixel 128*half + 4*ks + kk][o]
  float gb[GLDS ? 1 : 32];
  const float* g_src = G;
  int g_y0 = 0, g_x0 = 0;
  // GLDS staging map: item t = (pixel t / 3 of the 8 x 32 tile, float4 t % 3 of its 12 channels); thread tid owns items
  // tid and tid + 512 (the second only for tid < 256)
  float* gl_base = smem + 2 * kHH * kHW * kPSW;   // [2][kGL]
  float4 gq[2];
  const float* gq_src[2] = {G, G};
  bool gq_ok[2] = {false, false};
  auto stage_begin = [&](int tile) {
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
    const int gx = tx * kTW - 1 + s_hx;
    s_y0 = ty * kTH - 1;
    s_col = gx >= 0 && gx < W;
    s_src = Z + ((size_t)b * H * W + min(max(gx, 0), W - 1)) * 48 + 4 * s_q;
    g_y0 = ty * kTH + 4 * half;
    g_x0 = tx * kTW + kk;
    g_src = G + (size_t)b * H * W * ldg + c0 + min(r, 11);
    if constexpr (GLDS) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = min(tid + 512 * j, 767), pix = t / 3, piece = t - 3 * pix;
        const int gy = ty * kTH + (pix >> 5), gx = tx * kTW + (pix & 31);
        gq_ok[j] = gy < H && gx < W;
        gq_src[j] = G + ((size_t)(b * H + min(gy, H - 1)) * W + min(gx, W - 1)) * ldg + c0 + 4 * piece;
      }
    }
  };
  auto gq_load = [&](int j) { gq[j] = *reinterpret_cast<const float4*>(gq_src[j]); };   // unconditional, clamped
  auto gq_commit = [&](int j, float* dst) {   // pixels outside the image contribute zero
    const int t = min(tid + 512 * j, 767);
    const float4 v = gq_ok[j] ? gq[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (j == 0 || tid < 256) *reinterpret_cast<float4*>(dst + 4 * t) = v;
  };
  auto stage_load = [&](int it) {
    zt[it] = *reinterpret_cast<const float4*>(s_src + (size_t)min(max(s_y0 + it, 0), H - 1) * W * 48);
  };
  auto stage_commit = [&](int it, float* tile_l) {
    const bool ok = s_col && s_y0 + it >= 0 && s_y0 + it < H;
    float4 v;
    v.x = ok ? fmaf(zt[it].x, sc.x, sh.x) : 0.f;
    v.y = ok ? fmaf(zt[it].y, sc.y, sh.y) : 0.f;
    v.z = ok ? fmaf(zt[it].z, sc.z, sh.z) : 0.f;
    v.w = ok ? fmaf(zt[it].w, sc.w, sh.w) : 0.f;
    *reinterpret_cast<float4*>(tile_l + s_dst + it * kHW * kPSW) = v;  // threads >= 408 duplicate column 33
  };
  auto g_load = [&](int ks) {  // unconditional, clamped; masked at use
    const int gy = min(g_y0 + (ks >> 3), H - 1), gx = min(g_x0 + 4 * (ks & 7), W - 1);
    gb[ks] = g_src[((size_t)gy * W + gx) * ldg];
  };

  int tile = blockIdx.x, cur = 0;
  if (tile < ntiles) {
    stage_begin(tile);
#pragma unroll
    for (int it = 0; it < kHH; ++it) stage_load(it);
    if constexpr (GLDS) {
      gq_load(0);
      gq_load(1);
      gq_commit(0, gl_base);
      gq_commit(1, gl_base);
    } else {
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) g_load(ks);
    }
#pragma unroll
    for (int it = 0; it < kHH; ++it) stage_commit(it, smem);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    // validity of THIS tile's g pixels (the registers hold clamped loads)
    const int cy0 = g_y0, cx0 = g_x0;
    const int nxt = tile + gridDim.x;
    stage_begin(nxt < ntiles ? nxt : tile);
    const float* tile_l = smem + cur * (kHH * kHW * kPSW);
    float* tile_n = smem + (cur ^ 1) * (kHH * kHW * kPSW);
    const float* gl_c = gl_base + cur * kGL + (128 * half + kk) * 12 + min(r, 11);
    float* gl_n = gl_base + (cur ^ 1) * kGL;
    float av[2][7];  // A fragments are read one k-step ahead of their MFMAs (the fences below pin this order)
    float gvl[2] = {0.f, 0.f};   // GLDS: so is the g operand -- read with the CURRENT step it made every MFMA group wait
                                 // for the prefetch issued in front of it (lgkmcnt(0))
#pragma unroll
    for (int i = 0; i < 7; ++i) av[0][i] = tile_l[aoff[i]];
    if constexpr (GLDS) gvl[0] = gl_c[0];
#ifndef EML_WX   // experiment builds (-DEML_WX=<bits>): 1 = no staging of the next tile (wrong results), 2 = no barrier,
#define EML_WX 0  // 4 = no scheduling fences in the k-loop, 8 = reads interleaved with the MFMAs by a sched_group_barrier pipeline
#endif
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      if (!(EML_WX & 1) && ks < kHH) stage_load(ks);
      if (!(EML_WX & 1) && GLDS && ks >= kHH && ks < kHH + 2) gq_load(ks - kHH);
      if (ks + 1 < 32) {
        const float* base = tile_l + (((ks + 1) >> 3) * kHW + 4 * ((ks + 1) & 7)) * kPSW;
#pragma unroll
        for (int i = 0; i < 7; ++i) av[(ks + 1) & 1][i] = base[aoff[i]];
        if constexpr (GLDS) gvl[(ks + 1) & 1] = gl_c[4 * (ks + 1) * 12];
      }
      if (!(EML_WX & 12)) __builtin_amdgcn_sched_barrier(0);
      float gv;
      if constexpr (GLDS) {
        gv = r < 12 ? gvl[ks & 1] : 0.f;   // lanes 12..15 of a row: the unused MFMA columns
      } else {
        const bool gok = r < 12 && cy0 + (ks >> 3) < H && cx0 + 4 * (ks & 7) < W;
        gv = gok ? gb[ks] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = mfma16(av[ks & 1][i], gv, acc[i]);
      if constexpr (!GLDS) g_load(ks);  // next tile's value, in place
      if (!(EML_WX & 1) && GLDS && ks >= 32 - kHH - 2 && ks < 32 - kHH) gq_commit(ks - (32 - kHH - 2), gl_n);
      if (!(EML_WX & 1) && ks >= 32 - kHH) stage_commit(ks - (32 - kHH), tile_n);
      if (EML_WX & 8) {   // experiment: the next step's LDS reads one by one in the shadow of this step's MFMAs (no burst)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (!(EML_WX & 4)) __builtin_amdgcn_sched_barrier(0);
    }
    if (!(EML_WX & 2)) eml::lds_barrier();
    if (!(EML_WX & 1)) cur ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int idx = w4 + 4 * i;
    if (idx < 27) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        partial[((((size_t)blockIdx.x * 2 + half) * 27 + idx) * 16 + 4 * kk + g) * 16 + r] = acc[i][g];
    }
  }
}


// =============================================================================== conv3x3 backward: data + weight, one pass
// Round 4: conv3x3_bwd_data is bound by HBM (4.0 TB/s, 43 % of it written) with the matrix pipe half
// idle; conv3x3_bwd_weight is bound by the matrix pipe and then re-reads the g and z tiles the data gradient had just
// staged (24.8 GB per step).  Here one 512-thread workgroup does both for a tile: the g halo tile (double-buffered) feeds
// the data gradient (K = 9 x 12) and, through its own-pixel rows, the weight gradient's k = pixel operand; the BN2(z)
// halo tile -- the weight gradient's other operand -- is read ONCE (10 x 34 pixels, +33 % over the tile's own z, which the
// statistics still take from HBM/L2 as raw z) into a single LDS buffer, refilled for the next tile from registers between
// two barriers.  The data gradient's 81 weight fragments live in LDS (they were 81 VGPRs); the weight gradient's
// accumulators persist across tiles exactly as in conv3x3_bwd_weight_kernel (same deal of the 27 (tap, 16-channel) tiles
// to waves and halves, same tile order per workgroup -> bitwise the same partials for the same grid).
#ifdef EML_STAMPS   // experiment build (tools/exp_build.sh stamps -DEML_STAMPS): shader-clock cycles wave 0 of every workgroup spends
// in each part of a tile, summed over tiles and workgroups: [0] phase A, [1] barrier 2, [2] phase B, [3] barrier 1, [4] tiles
__device__ unsigned long long eml_c3_stamps[8];
#define EML_C3_STAMP(slot)                                                   \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();            \
    st_acc[slot] += now_ - st_last;                                          \
    st_last = now_;                                                          \
  } while (0)
#else
#define EML_C3_STAMP(slot) do {} while (0)
#endif
// Needs the fused BN1 affine (X given).  A16: the 12-channel slices of G and X are 16-byte aligned (blocks 1 and 2); otherwise
// (block 3 of EMLight's encoder starts at channel 150) each staged item is fetched as two 8-byte loads.
template <bool A16>
__global__ __launch_bounds__(512, 2) void conv3x3_bwd_fused_kernel(
    const float* __restrict__ G, int ldg, int c0, const float* __restrict__ W2, const float* __restrict__ Z,
    const float* __restrict__ zmean, const float* __restrict__ zistd, float* __restrict__ DZ, int B, int H, int W,
    double* __restrict__ partials /*[grid][48][2]*/, const float* __restrict__ Xb, int ldx, int cx,
    const float* __restrict__ sB, const float* __restrict__ sC, float* __restrict__ GF,
    const float* __restrict__ scale2, const float* __restrict__ shift2, float* __restrict__ partialW /*[2*grid][27][16][16]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kGT = kHH * kHW * kPSG;                 // floats of a g halo tile
  float* g_l = smem;                                    // [2][kGT]
  float* z_l = g_l + 2 * kGT;                           // [kHH * kHW][kPSW]   BN2(z) halo tile, zero outside the image
  float* w_l = z_l + kHH * kHW * kPSW;                  // [27][3][64]         data-gradient A fragments
  float* coef_l = w_l + 27 * 3 * 64;                    // [24 (+8)]           sB | sC of the layer's 12 channels
  double* red = reinterpret_cast<double*>(coef_l + 32); // [8][48][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int half = wave >> 2, w4 = wave & 3;

  // data-gradient fragments (D^T form): entry ((tap*3 + s)*3 + n)*64 + (kk*16 + r) = W2[o = 4s + kk][c = 16n + r][tap]
  for (int e = tid; e < 27 * 3 * 64; e += 512) {
    const int l = e & 63, g3 = e >> 6, n = g3 % 3, ts = g3 / 3, s3 = ts % 3, tap = ts / 3;
    w_l[e] = W2[((size_t)(4 * s3 + (l >> 4)) * 48 + 16 * n + (l & 15)) * 9 + tap];
  }
  if (tid < 24) coef_l[tid] = tid < 12 ? sB[cx + tid] : sC[cx + tid - 12];

  // weight-gradient accumulators and operand offsets (conv3x3_bwd_weight_kernel)
  f32x4 accw[7];
  int aoff[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    accw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int idx = min(w4 + 4 * i, 26), tap = idx / 3, mc = idx - 3 * tap;
    aoff[i] = ((tap / 3 + 4 * half) * kHW + (tap % 3) + kk) * kPSW + 16 * mc + r;
  }
  const int tx_n = (W + kTW - 1) / kTW, ty_n = (H + kTH - 1) / kTH;
  const int ntiles = B * ty_n * tx_n;
  // BatchNorm statistics: f64 running sums per (wave, channel) in LDS (`red`), updated once per tile by the lane that owns
  // the channel after a DPP reduction over the 16 pixel lanes -- as 48 VGPRs of doubles they pushed the kernel into spills
  for (int e = tid; e < 8 * 48 * 2; e += 512) red[e] = 0.0;

  // ---- g halo staging (the WIDE map of conv3x3_bwd_data_kernel): item t = tid + 512*it, halo pixel t / 3, float4 t % 3
  constexpr int kWideItems = kHH * kHW * 3;
  int w_hy[2], w_hx[2], w_q[2], w_pix[2];
  bool w_ok[2], w_own[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = min(tid + 512 * it, kWideItems - 1), hp = t / 3;
    w_q[it] = t - 3 * hp;
    w_hy[it] = hp / kHW;
    w_hx[it] = hp - w_hy[it] * kHW;
  }
  float4 gt4[2], xt4[2];
  // ---- z halo staging (conv3x3_bwd_weight_kernel): thread = (halo column, 16-byte slice), one halo row per pass
  const int s_hx = min(tid / 12, kHW - 1), s_q = tid % 12;
  const float4 sc = *reinterpret_cast<const float4*>(scale2 + 4 * s_q);
  const float4 sh = *reinterpret_cast<const float4*>(shift2 + 4 * s_q);
  const int s_dst = s_hx * kPSW + 4 * s_q;
  float4 zt[kHH];
  const float* s_src = Z;
  int s_y0 = 0;
  bool s_col = false;
  auto stage_begin = [&](int tile) {
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int gy = ty * kTH - 1 + w_hy[it], gx = tx * kTW - 1 + w_hx[it];
      w_ok[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
      w_pix[it] = (b * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1);
      w_own[it] = w_ok[it] && w_hy[it] >= 1 && w_hy[it] <= kTH && w_hx[it] >= 1 && w_hx[it] <= kTW;
    }
    const int gx = tx * kTW - 1 + s_hx;
    s_y0 = ty * kTH - 1;
    s_col = gx >= 0 && gx < W;
    s_src = Z + ((size_t)b * H * W + min(max(gx, 0), W - 1)) * 48 + 4 * s_q;
  };
  auto g_load = [&](int it) {   // unconditional, clamped
    const float* gp = G + (size_t)w_pix[it] * ldg + c0 + 4 * w_q[it];
    const float* xp = Xb + (size_t)w_pix[it] * ldx + cx + 4 * w_q[it];
    if constexpr (A16) {
      gt4[it] = *reinterpret_cast<const float4*>(gp);
      xt4[it] = *reinterpret_cast<const float4*>(xp);
    } else {
      const float2 g0 = *reinterpret_cast<const float2*>(gp), g1 = *reinterpret_cast<const float2*>(gp + 2);
      const float2 x0 = *reinterpret_cast<const float2*>(xp), x1 = *reinterpret_cast<const float2*>(xp + 2);
      gt4[it] = make_float4(g0.x, g0.y, g1.x, g1.y);
      xt4[it] = make_float4(x0.x, x0.y, x1.x, x1.y);
    }
  };
  auto g_commit = [&](int it, float* dst) {
    const float4 fb4 = *reinterpret_cast<const float4*>(coef_l + 4 * w_q[it]);
    const float4 fc4 = *reinterpret_cast<const float4*>(coef_l + 12 + 4 * w_q[it]);
    float4 v;
    v.x = fmaf(fb4.x, xt4[it].x, gt4[it].x) + fc4.x;
    v.y = fmaf(fb4.y, xt4[it].y, gt4[it].y) + fc4.y;
    v.z = fmaf(fb4.z, xt4[it].z, gt4[it].z) + fc4.z;
    v.w = fmaf(fb4.w, xt4[it].w, gt4[it].w) + fc4.w;
    if (!w_ok[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* d = dst + (w_hy[it] * kHW + w_hx[it]) * kPSG + 4 * w_q[it];   // 56-byte pixel stride: 8-byte aligned
    *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    if (w_own[it]) *reinterpret_cast<float4*>(GF + (size_t)w_pix[it] * 12 + 4 * w_q[it]) = v;
  };
  auto z_load = [&](int it) {
    zt[it] = *reinterpret_cast<const float4*>(s_src + (size_t)min(max(s_y0 + it, 0), H - 1) * W * 48);
  };
  auto z_commit = [&](int it, int y0, bool col) {
    const bool ok = col && y0 + it >= 0 && y0 + it < H;
    float4 v;
    v.x = ok ? fmaf(zt[it].x, sc.x, sh.x) : 0.f;
    v.y = ok ? fmaf(zt[it].y, sc.y, sh.y) : 0.f;
    v.z = ok ? fmaf(zt[it].z, sc.z, sh.z) : 0.f;
    v.w = ok ? fmaf(zt[it].w, sc.w, sh.w) : 0.f;
    *reinterpret_cast<float4*>(z_l + s_dst + it * kHW * kPSW) = v;   // threads >= 408 duplicate column 33
  };

  int tile = blockIdx.x, cur = 0;
  __syncthreads();   // coef_l, w_l, red
  if (tile < ntiles) {
    stage_begin(tile);
    g_load(0);
    g_load(1);
#pragma unroll
    for (int it = 0; it < kHH; ++it) z_load(it);   // committed by the first phase A, like every later tile's
    g_commit(0, g_l);
    g_commit(1, g_l);
  }
  __syncthreads();
  // Per tile: phase A (data gradient) with the z halo's LDS commit riding on it; barrier; phase B (weight gradient) with
  // the data gradient's epilogue (statistics, dzn stores) and the next tile's z halo loads riding on it; barrier.  Nothing
  // but MFMA-fed work sits between the barriers: with one workgroup per CU the two waves of a SIMD share every phase, so
  // any serial section (the epilogue was ~150 VALU + 6 stores per lane, the z commit 10 ds_write_b128) idled the matrix pipe.
#ifdef EML_STAMPS
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#endif
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const float* gc = g_l + cur * kGT;
    float* gn = g_l + (cur ^ 1) * kGT;
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
    const int gy = ty * kTH + wave;
    size_t prow[2];
    bool pv[2];
    float4 zr[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int gx = tx * kTW + 16 * m + r;
      pv[m] = gy < H && gx < W;
      prow[m] = ((size_t)(b * H + min(gy, H - 1)) * W + min(gx, W - 1)) * 48;
    }
    // validity of THIS tile's z halo rows / column (stage_begin below moves on to the next tile)
    const int cz_y0 = s_y0;
    const bool cz_col = s_col;
    stage_begin(nxt < ntiles ? nxt : tile);   // (the last tile re-stages itself: loads stay unconditional)

    // ------------------------------------------------------------------ phase A: data gradient (162 MFMAs per wave)
    f32x4 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float aq[2][2], wq[2][3];
#pragma unroll
    for (int n = 0; n < 3; ++n) wq[0][n] = w_l[n * 64 + lane];
#pragma unroll
    for (int m = 0; m < 2; ++m) aq[0][m] = gc[((wave + 2) * kHW + 16 * m + r + 2) * kPSG + kk];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int gi = tap * 3 + s3;
        if (gi < 2) {
          g_load(gi);
          __builtin_amdgcn_sched_barrier(0);
        } else if (gi < 2 + kHH) {
          z_commit(gi - 2, cz_y0, cz_col);   // this tile's BN2(z) halo row: nobody reads z_l before the barrier below
          __builtin_amdgcn_sched_barrier(0);
        }
        // operands of group gi + 1 are requested before the MFMAs of group gi (read right in front of their MFMAs every
        // group of 6 waited out an LDS round trip)
        if (gi + 1 < 27) {
          const int tn = (gi + 1) / 3, sn = (gi + 1) - 3 * tn, dyn = tn / 3, dxn = tn - 3 * dyn;
#pragma unroll
          for (int n = 0; n < 3; ++n) wq[(gi + 1) & 1][n] = w_l[((gi + 1) * 3 + n) * 64 + lane];
#pragma unroll
          for (int m = 0; m < 2; ++m)
            aq[(gi + 1) & 1][m] = gc[((wave + 2 - dyn) * kHW + 16 * m + r + 2 - dxn) * kPSG + 4 * sn + kk];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[m][n] = mfma16(wq[gi & 1][n], aq[gi & 1][m], acc[m][n]);  // D[channel][pixel]
        if (gi >= 25) {
          __builtin_amdgcn_sched_barrier(0);
          g_commit(gi - 25, gn);
        }
      }
    }
    EML_C3_STAMP(0);
    eml::lds_barrier();          // this tile's z halo is in place (and g_l[cur ^ 1] complete)
    EML_C3_STAMP(1);
    // ------------------------------------------------------------------ phase B: weight gradient (216 MFMAs per wave)
    // k = pixel: lane (kk, o = r) takes g of own pixel 128*half + 4*ks + kk from the g halo tile (halo row / column + 1);
    // the next tile's z halo rows are requested under the first 10 k-steps and committed after the barrier below
    const float* gl_c = gc + ((4 * half + 1) * kHW + kk + 1) * kPSG + min(r, 11);
    float av[2][7], gvl[2];   // both operands are read one k-step ahead of their MFMAs
#pragma unroll
    for (int i = 0; i < 7; ++i) av[0][i] = z_l[aoff[i]];
    gvl[0] = gl_c[0];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      // requests riding on this phase: the tile's own raw z rows (statistics, k-steps 0..5), then the NEXT tile's z halo
      if (ks < 6) zr[ks / 3][ks % 3] = *reinterpret_cast<const float4*>(Z + prow[ks / 3] + 16 * (ks % 3) + 4 * kk);
      else if (ks < 6 + kHH) z_load(ks - 6);
      if (ks + 1 < 32) {
        const float* base = z_l + (((ks + 1) >> 3) * kHW + 4 * ((ks + 1) & 7)) * kPSW;
#pragma unroll
        for (int i = 0; i < 7; ++i) av[(ks + 1) & 1][i] = base[aoff[i]];
        gvl[(ks + 1) & 1] = gl_c[(((ks + 1) >> 3) * kHW + 4 * ((ks + 1) & 7)) * kPSG];
      }
      __builtin_amdgcn_sched_barrier(0);
      const float gv = r < 12 ? gvl[ks & 1] : 0.f;   // (zero outside the image: staged so)
#pragma unroll
      for (int i = 0; i < 7; ++i) accw[i] = mfma16(av[ks & 1][i], gv, accw[i]);
      // the data gradient's epilogue, piecewise behind these MFMAs: statistics of channel group n at k-step 16 + n (its z
      // rows were requested ten k-steps ago), then one 16-byte dzn store per k-step
      if (ks >= 6 + kHH && ks < 9 + kHH) {
        const int n = ks - (6 + kHH);
        float l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float4 z = zr[m][n];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float v = pv[m] ? acc[m][n][g] : 0.f;
            l1[g] += v;
            l2[g] = fmaf(v, f4c(z, g), l2[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float t1 = eml::row16_sum(l1[g]), t2 = eml::row16_sum(l2[g]);   // over this tile row's 32 pixels
          if (r == 0) {   // wave-private slots: no other lane touches them
            double* d = red + (wave * 48 + 16 * n + 4 * kk + g) * 2;
            d[0] += (double)t1;
            d[1] += (double)t2;
          }
        }
      } else if (ks >= 9 + kHH && ks < 15 + kHH) {
        const int si = ks - (9 + kHH), m = si / 3, n = si - 3 * m;
        if (pv[m])
          *reinterpret_cast<float4*>(DZ + prow[m] + 16 * n + 4 * kk) =
              make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    EML_C3_STAMP(2);
    eml::lds_barrier();          // everyone is done with z_l and g_l[cur]
    EML_C3_STAMP(3);
#ifdef EML_STAMPS
    st_acc[4] += 1;
#endif
    cur ^= 1;
  }
#ifdef EML_STAMPS
  if (tid == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&eml_c3_stamps[i], st_acc[i]);
#endif
  // ---- outputs: weight-gradient partials per (workgroup, half), then the BatchNorm statistics
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int idx = w4 + 4 * i;
    if (idx < 27) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        partialW[((((size_t)blockIdx.x * 2 + half) * 27 + idx) * 16 + 4 * kk + g) * 16 + r] = accw[i][g];
    }
  }
  __syncthreads();
  if (tid < 48) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) {
      t1 += red[(w8 * 48 + tid) * 2 + 0];
      t2 += red[(w8 * 48 + tid) * 2 + 1];
    }
    partials[(size_t)blockIdx.x * 96 + 2 * tid + 0] = t1;
    partials[(size_t)blockIdx.x * 96 + 2 * tid + 1] = (double)zistd[tid] * (t2 - (double)zmean[tid] * t1);
  }
}


// =============================================================================== conv3x3 backward, one pass, weight gradient TAP-PACKED
// Round 6.  conv3x3_bwd_fused_kernel's weight gradient deals the 27 (tap, 16 input channels) accumulator tiles of
//     dW2[o][c][tap] = sum_p g[p][o] * zn[p + d_tap][c]
// to 2 halves x 4 waves with the 12 output channels as the MFMA columns (12 of 16 used) and one of the 28 slots empty: 224
// MFMAs per wave and 256-pixel tile, and its operand is the BN2(z) HALO tile (10 x 34 pixels, each read 9 times).  Here the
// sum runs over the pixel p' = p + d that zn is taken at:
//     dW2[(tap, o)][c] = sum_{p'} g[p' - d_tap][o] * zn[p'][c]
// rows (tap, o) = 108 of 112 in 7 row tiles, columns c = 48 in 3: 21 accumulator tiles instead of 27; every wave keeps all 21
// (84 registers) and takes the k-steps of its OWN tile row (32 pixels = 8 k-steps): 168 MFMAs per wave and tile, balanced,
// against 224 (-25 %; the kernel: 330 against 386, -14.5 %).  The operand that needs neighbours is now g, whose halo tile the
// data gradient stages anyway (a per-lane offset into it per row tile: ds_read_b32, as before); z is needed at the tile's own
// 256 pixels only and is staged RAW (no halo: -25 % of the z loads and LDS writes): BN2's affine is one fma on the fragment, and
// the data gradient's BatchNorm statistics read the same raw rows from LDS instead of a second time from L2 (6 x 16-byte
// global loads per lane and tile gone).  The eight waves' accumulators are summed through LDS once, after the last tile (fixed
// order): one partial row of 21 x 256 floats per workgroup (it was 2 x 27 x 256).  Data gradient, g staging and the phase
// structure (A: data gradient, barrier, B: weight gradient with the data gradient's epilogue riding on it, barrier) are
// conv3x3_bwd_fused_kernel's.
template <bool A16>
__global__ __launch_bounds__(512, 2) void conv3x3_bwd_fused_tp_kernel(
    const float* __restrict__ G, int ldg, int c0, const float* __restrict__ W2, const float* __restrict__ Z,
    const float* __restrict__ zmean, const float* __restrict__ zistd, float* __restrict__ DZ, int B, int H, int W,
    double* __restrict__ partials /*[grid][48][2]*/, const float* __restrict__ Xb, int ldx, int cx,
    const float* __restrict__ sB, const float* __restrict__ sC, float* __restrict__ GF,
    const float* __restrict__ scale2, const float* __restrict__ shift2, float* __restrict__ partialW /*[grid][21][16][16]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kGT = kHH * kHW * kPSG;                 // floats of a g halo tile
  constexpr int kZS = 48;                               // pixel stride of the raw z tile (ds_read_b32 over 4 pixels x 16 channels)
  float* g_l = smem;                                    // [2][kGT]
  float* z_l = g_l + 2 * kGT;                           // [2][kTH * kTW][kZS] raw z of the tile's own pixels, double-buffered
  float* w_l = z_l + 2 * kTH * kTW * kZS;               // [27][3][64]         data-gradient A fragments
  float* coef_l = w_l + 27 * 3 * 64;                    // [24 (+8)]           sB | sC of the layer's 12 channels
  double* red = reinterpret_cast<double*>(coef_l + 32); // [8][48][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;

  // data-gradient fragments (D^T form): entry ((tap*3 + s)*3 + n)*64 + (kk*16 + r) = W2[o = 4s + kk][c = 16n + r][tap]
  for (int e = tid; e < 27 * 3 * 64; e += 512) {
    const int l = e & 63, g3 = e >> 6, n = g3 % 3, ts = g3 / 3, s3 = ts % 3, tap = ts / 3;
    w_l[e] = W2[((size_t)(4 * s3 + (l >> 4)) * 48 + 16 * n + (l & 15)) * 9 + tap];
  }
  if (tid < 24) coef_l[tid] = tid < 12 ? sB[cx + tid] : sC[cx + tid - 12];

  // weight gradient: accumulator tile (t, n) = rows 16t .. 16t + 15 of (tap, o) = (row / 12, row % 12), columns 16n .. + 15 of c.
  // Lane (i = r, k = kk) of an A fragment reads g of halo pixel (own pixel k) - d_tap, channel o: a per-lane offset per row tile.
  // Waves j and j + 4 (the two waves of SIMD j) share the 64 pixels of tile rows 2j, 2j + 1 (16 k-steps): wave j keeps row
  // tiles 0..2 (9 accumulator tiles), wave j + 4 row tiles 3..6 (12): 21 tiles x 16 k-steps of MFMAs per SIMD, and 48
  // accumulator registers instead of the 84 of "every wave keeps all 21 tiles of its own row" (which spilled: every reload in
  // phase A was an s_waitcnt vmcnt(0) on the z DMA in flight, profiles/r06_c3bwd_tp_stamps.txt).
  const int role = __builtin_amdgcn_readfirstlane(wave >> 2), pairj = wave & 3;
  f32x4 accw[4][3];
  int goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int n = 0; n < 3; ++n) accw[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t = min(3 * role + i, 6);
    const int row = min(16 * t + r, 107), tap = row / 12, o = row - 12 * tap;
    goff[i] = (-(tap / 3 - 1) * kHW - (tap % 3 - 1)) * kPSG + o;
  }
  float scb[3], shb[3];   // BN2 of the B fragment's channel 16n + r
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    scb[n] = scale2[16 * n + r];
    shb[n] = shift2[16 * n + r];
  }
  const int tx_n = (W + kTW - 1) / kTW, ty_n = (H + kTH - 1) / kTH;
  const int ntiles = B * ty_n * tx_n;
  for (int e = tid; e < 8 * 48 * 2; e += 512) red[e] = 0.0;

  // ---- g halo staging (the WIDE map of conv3x3_bwd_data_kernel): item t = tid + 512*it, halo pixel t / 3, float4 t % 3
  constexpr int kWideItems = kHH * kHW * 3;
  int w_hy[2], w_hx[2], w_q[2], w_pix[2];
  bool w_ok[2], w_own[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = min(tid + 512 * it, kWideItems - 1), hp = t / 3;
    w_q[it] = t - 3 * hp;
    w_hy[it] = hp / kHW;
    w_hx[it] = hp - w_hy[it] * kHW;
  }
  f32x4 gt4[2], xt4[2];   // (the native vector type: written by asm statements, see g_load)
  // ---- raw z of the own pixels: 256 pixels x 12 float4 = 6 items per thread, item t = tid + 512*it = pixel t / 12, slice t % 12
  // (consecutive threads walk a pixel's 192 bytes, then the next pixel of the tile row: whole cache lines)
  constexpr int kZI = kTH * kTW * 12 / 512;   // 6
  int z_pq[kZI];                              // pixel << 4 | slice
#pragma unroll
  for (int it = 0; it < kZI; ++it) {
    const int t = tid + 512 * it;
    z_pq[it] = ((t / 12) << 4) | (t % 12);
  }
  auto stage_begin = [&](int tile) {
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int gy = ty * kTH - 1 + w_hy[it], gx = tx * kTW - 1 + w_hx[it];
      w_ok[it] = gy >= 0 && gy < H && gx >= 0 && gx < W;
      w_pix[it] = (b * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1);
      w_own[it] = w_ok[it] && w_hy[it] >= 1 && w_hy[it] <= kTH && w_hx[it] >= 1 && w_hx[it] <= kTW;
    }
  };
  auto g_load = [&](int it) {   // unconditional, clamped
    const float* gp = G + (size_t)w_pix[it] * ldg + c0 + 4 * w_q[it];
    const float* xp = Xb + (size_t)w_pix[it] * ldx + cx + 4 * w_q[it];
    if constexpr (A16) {
      // through asm statements: as ordinary loads from read-only memory the second pair was REMATERIALISED by the register
      // allocator -- issued again at MFMA 146 of the phase's 162, an HBM round trip in front of every tile's barrier (the
      // first copy, at the top of the phase, deleted).  The compiler keeps no count of these: g_wait() before the commits.
      // Straight into the variables the commit reads, and nothing touches them before g_wait(): a COPY made while the load is
      // in flight moves the register's old contents (the first version copied into a float4 struct: NaNs in the step).
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(gt4[it]) : "v"(gp) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xt4[it]) : "v"(xp) : "memory");
    } else {
      const float2 g0 = *reinterpret_cast<const float2*>(gp), g1 = *reinterpret_cast<const float2*>(gp + 2);
      const float2 x0 = *reinterpret_cast<const float2*>(xp), x1 = *reinterpret_cast<const float2*>(xp + 2);
      gt4[it] = f32x4{g0.x, g0.y, g1.x, g1.y};
      xt4[it] = f32x4{x0.x, x0.y, x1.x, x1.y};
    }
  };
  // the LDS half of a commit; the compact GF copy of the tile's own pixels leaves separately (g_store): its exec-masked
  // store opens a new basic block, and with the second item's commit behind the first item's store the compiler SANK the second
  // pair of staging loads from the top of phase A to its end (MachineSink: the values are only used in that later block) --
  // an HBM round trip exposed in front of the barrier of every tile
  auto g_wait = [&]() {   // the staging loads have landed (ties the registers to the wait: nothing is read before it)
    if constexpr (A16)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(gt4[0]), "+v"(xt4[0]), "+v"(gt4[1]), "+v"(xt4[1]) :: "memory");
  };
  auto g_commit = [&](int it, float* dst) {
    const float4 fb4 = *reinterpret_cast<const float4*>(coef_l + 4 * w_q[it]);
    const float4 fc4 = *reinterpret_cast<const float4*>(coef_l + 12 + 4 * w_q[it]);
    float4 v;
    v.x = fmaf(fb4.x, xt4[it][0], gt4[it][0]) + fc4.x;
    v.y = fmaf(fb4.y, xt4[it][1], gt4[it][1]) + fc4.y;
    v.z = fmaf(fb4.z, xt4[it][2], gt4[it][2]) + fc4.z;
    v.w = fmaf(fb4.w, xt4[it][3], gt4[it][3]) + fc4.w;
    if (!w_ok[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* d = dst + (w_hy[it] * kHW + w_hx[it]) * kPSG + 4 * w_q[it];   // 56-byte pixel stride: 8-byte aligned
    *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    gt4[it] = f32x4{v.x, v.y, v.z, v.w};   // (kept for g_store)
  };
  auto g_store = [&](int it) {
    // 32-bit element offset from the uniform base (P * 12 < 2^31): as a 64-bit per-lane pointer its loop-invariant part was
    // spilled and reloaded (s_waitcnt vmcnt(0)) in front of every store
    if (w_own[it]) *reinterpret_cast<float4*>(GF + ((unsigned)w_pix[it] * 12u + 4u * (unsigned)w_q[it])) = make_float4(gt4[it][0], gt4[it][1], gt4[it][2], gt4[it][3]);
  };
  // z goes HBM -> LDS by DMA (global_load_lds_dwordx4: 16 bytes per lane, lane-linear: item t lands at z_l + 4 t floats, which
  // IS [pixel][48] for t = 12 pixel + slice): no staging registers (held across a barrier they were spilled, each load with
  // its own vmcnt(0)), no ds_write.  The NEXT tile's z is requested at the top of a tile's phase B into the other buffer and
  // has a phase B, a barrier and a phase A to land (requested at the top of its own phase A it cost that phase 10 000 cycles
  // of waiting: profiles/r06_c3bwd_tp_stamps.txt); s_waitcnt vmcnt(0) in front of the barrier that ends phase A.  Through an asm statement
  // (cdna_hip_programming.md 5.7): the compiler keeps no record of it and neither drains the VM counter in front of LDS reads
  // nor reorders around the explicit wait.  Pixels outside the image are read from a clamped address and masked where the
  // fragments are built.
  auto z_dma = [&](int it, int b, int y0, int x0, int buf) {
    const int pix = z_pq[it] >> 4, q = z_pq[it] & 15;
    const int gy = min(y0 + (pix >> 5), H - 1), gx = min(x0 + (pix & 31), W - 1);
    const float* src = Z + ((size_t)(b * H + gy) * W + gx) * 48 + 4 * q;
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(z_l + buf * (kTH * kTW * kZS) + 4 * (64 * wave + 512 * it)));
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };

  int tile = blockIdx.x, cur = 0;
  __syncthreads();   // coef_l, w_l, red
  if (tile < ntiles) {
    stage_begin(tile);
    g_load(0);
    g_load(1);
    {
      const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
#pragma unroll
      for (int it = 0; it < kZI; ++it) z_dma(it, b, (rem / tx_n) * kTH, (rem % tx_n) * kTW, 0);
    }
    g_wait();
    g_commit(0, g_l);
    g_commit(1, g_l);
    g_store(0);
    g_store(1);
  }
  __syncthreads();
#ifdef EML_STAMPS
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#endif
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    const float* gc = g_l + cur * kGT;
    float* gn = g_l + (cur ^ 1) * kGT;
    const int b = tile / (ty_n * tx_n), rem = tile - b * (ty_n * tx_n);
    const int ty = rem / tx_n, tx = rem - ty * tx_n;
    const int gy = ty * kTH + wave;
    size_t prow[2];
    bool pv[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int gx = tx * kTW + 16 * m + r;
      pv[m] = gy < H && gx < W;
      prow[m] = ((size_t)(b * H + min(gy, H - 1)) * W + min(gx, W - 1)) * 48;
    }
    stage_begin(nxt < ntiles ? nxt : tile);   // (the last tile re-stages itself: loads stay unconditional)
    // ------------------------------------------------------------------ phase A: data gradient (162 MFMAs per wave)
    f32x4 acc[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float aq[2][2], wq[2][3];
#pragma unroll
    for (int n = 0; n < 3; ++n) wq[0][n] = w_l[n * 64 + lane];
#pragma unroll
    for (int m = 0; m < 2; ++m) aq[0][m] = gc[((wave + 2) * kHW + 16 * m + r + 2) * kPSG + kk];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int gi = tap * 3 + s3;
        if (gi < 2) {
          g_load(gi);
          // a compiler-level memory barrier: with no store left in this phase (round 4's kernel committed the z halo here) LLVM
          // sank the second pair of staging loads from here to their use at the end of the phase -- an HBM round trip in
          // front of every tile's barrier (ISA: global_load at MFMA 150 of 162, s_waitcnt vmcnt(0) at 156)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        if (gi + 1 < 27) {
          const int tn = (gi + 1) / 3, sn = (gi + 1) - 3 * tn, dyn = tn / 3, dxn = tn - 3 * dyn;
#pragma unroll
          for (int n = 0; n < 3; ++n) wq[(gi + 1) & 1][n] = w_l[((gi + 1) * 3 + n) * 64 + lane];
#pragma unroll
          for (int m = 0; m < 2; ++m)
            aq[(gi + 1) & 1][m] = gc[((wave + 2 - dyn) * kHW + 16 * m + r + 2 - dxn) * kPSG + 4 * sn + kk];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[m][n] = mfma16(wq[gi & 1][n], aq[gi & 1][m], acc[m][n]);  // D[channel][pixel]
        if (gi >= 25) {
          __builtin_amdgcn_sched_barrier(0);
          if (gi == 25) g_wait();
          g_commit(gi - 25, gn);
        }
      }
    }
    g_store(0);
    g_store(1);
    EML_C3_STAMP(0);
#ifndef EML_C3_NOZWAIT   // experiment build: what the wait costs (stale z: wrong results)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the z tile has landed
#endif
    eml::lds_barrier();          // this tile's z is in place (and g_l[cur ^ 1] complete)
    EML_C3_STAMP(1);
    // ------------------------------------------------------------------ phase B: weight gradient (144 / 192 MFMAs per wave; row tile 3 shared between the pair -- 168 each -- measured slower: 14 spills)
    // k = pixel x = 4 (ks % 8) + kk of tile row 2 pairj + ks / 8: A = g at that pixel minus the row's tap offset, B = BN2(z) there
    const float* ga = gc + ((2 * pairj + 1) * kHW + kk + 1) * kPSG;
    const float* zc = z_l + cur * (kTH * kTW * kZS);
    const float* zb = zc + (kTW * 2 * pairj + kk) * kZS + r;
    // B-fragment pixels of the pair's two tile rows that lie in the image: x = 4 (ks % 8) + kk < xl[ks / 8]
    const int xl0 = ty * kTH + 2 * pairj < H ? W - tx * kTW - kk : 0, xl1 = ty * kTH + 2 * pairj + 1 < H ? W - tx * kTW - kk : 0;
    auto phase_b = [&](auto nt_tag) {
      constexpr int NT = decltype(nt_tag)::value;   // row tiles of this wave: 3 (tiles 0..2) or 4 (tiles 3..6)
      float av[2][NT], bz[2][3];                    // both operands are read one k-step ahead of their MFMAs
#pragma unroll
      for (int i = 0; i < NT; ++i) av[0][i] = ga[goff[i]];
#pragma unroll
      for (int n = 0; n < 3; ++n) bz[0][n] = zb[16 * n];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + 1 < 16) {
          const int k1 = ks + 1, go = ((k1 >> 3) * kHW + 4 * (k1 & 7)) * kPSG, zo = ((k1 >> 3) * kTW + 4 * (k1 & 7)) * kZS;
#pragma unroll
          for (int i = 0; i < NT; ++i) av[k1 & 1][i] = ga[go + goff[i]];
#pragma unroll
          for (int n = 0; n < 3; ++n) bz[k1 & 1][n] = zb[zo + 16 * n];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 1) {   // the next tile's raw z -> the other buffer (the last tile re-requests itself: the loop stays branch-free).
                         // Behind the phase's address set-up: the compiler reloads a few spilled loop invariants there, each
                         // with s_waitcnt vmcnt(0) -- issued in front of them, every tile waited out its own request here.
          const int t2 = nxt < ntiles ? nxt : tile;
          const int b2 = t2 / (ty_n * tx_n), rem2 = t2 - b2 * (ty_n * tx_n);
#ifndef EML_C3_NOZDMA   // experiment build: no z traffic at all (wrong results)
#pragma unroll
          for (int it = 0; it < kZI; ++it) z_dma(it, b2, (rem2 / tx_n) * kTH, (rem2 % tx_n) * kTW, cur ^ 1);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        const bool pin = 4 * (ks & 7) < (ks < 8 ? xl0 : xl1);
        float bv[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) bv[n] = pin ? fmaf(bz[ks & 1][n], scb[n], shb[n]) : 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const float a = (i < 3 || r < 12) ? av[ks & 1][i] : 0.f;   // rows 108..111 of row tile 6 do not exist
#pragma unroll
          for (int n = 0; n < 3; ++n) accw[i][n] = mfma16(a, bv[n], accw[i][n]);
        }
        // the data gradient's epilogue (this wave's OWN tile row `wave`), piecewise behind these MFMAs: statistics of channel
        // group n at k-steps 2, 4, 6 (raw z rows from LDS), then one 16-byte dzn store per k-step
        if (ks == 2 || ks == 4 || ks == 6) {
          const int n = (ks - 2) >> 1;
          float l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const float4 z = *reinterpret_cast<const float4*>(zc + (kTW * wave + 16 * m + r) * kZS + 16 * n + 4 * kk);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float v = pv[m] ? acc[m][n][g] : 0.f;
              l1[g] += v;
              l2[g] = fmaf(v, f4c(z, g), l2[g]);
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float t1 = eml::row16_sum(l1[g]), t2 = eml::row16_sum(l2[g]);   // over this tile row's 32 pixels
            if (r == 0) {   // wave-private slots: no other lane touches them
              double* d = red + (wave * 48 + 16 * n + 4 * kk + g) * 2;
              d[0] += (double)t1;
              d[1] += (double)t2;
            }
          }
        } else if (ks >= 8 && ks < 14) {
          const int si = ks - 8, m = si / 3, n = si - 3 * m;
          if (pv[m])
            *reinterpret_cast<float4*>(DZ + prow[m] + 16 * n + 4 * kk) =
                make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (role == 0) phase_b(std::integral_constant<int, 3>{});
    else phase_b(std::integral_constant<int, 4>{});
    EML_C3_STAMP(2);
    eml::lds_barrier();          // everyone is done with z_l[cur] and g_l[cur]
    EML_C3_STAMP(3);
#ifdef EML_STAMPS
    st_acc[4] += 1;
#endif
    cur ^= 1;
  }
#ifdef EML_STAMPS
  if (tid == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(&eml_c3_stamps[i], st_acc[i]);
#endif
  // ---- outputs: the eight waves' weight-gradient accumulators summed through LDS in wave order, one partial row per workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last tile's look-ahead request still writes z_l)
  __syncthreads();
  float* wsum = smem;            // [21][16][16] (the g / z tiles are dead)
  for (int w8 = 0; w8 < 8; ++w8) {   // waves 0..3 hold row tiles 0..2, waves 4..7 row tiles 3..6: four contributions per tile
    if (wave == w8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < 3 + role) {
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float* q = wsum + ((((3 * role + i) * 3 + n) * 16 + 4 * kk + g) * 16 + r);
              *q = ((w8 & 3) == 0 ? 0.f : *q) + accw[i][n][g];
            }
        }
    }
    __syncthreads();
  }
  for (int e = tid; e < 21 * 256; e += 512) partialW[(size_t)blockIdx.x * (21 * 256) + e] = wsum[e];
  if (tid < 48) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) {
      t1 += red[(w8 * 48 + tid) * 2 + 0];
      t2 += red[(w8 * 48 + tid) * 2 + 1];
    }
    partials[(size_t)blockIdx.x * 96 + 2 * tid + 0] = t1;
    partials[(size_t)blockIdx.x * 96 + 2 * tid + 1] = (double)zistd[tid] * (t2 - (double)zmean[tid] * t1);
  }
}


// =============================================================================== BN backward: finalize
// From the partial (S1 = sum dy, S2 = sum dy*xhat) of C channels: dgamma = S2, dbeta = S1 and the
// per-channel affine of the input gradient  dx = cA*dy + cB*x + cC  where
//   cA = gamma*istd,  cB = -gamma*istd^2*S2/n,  cC = -gamma*istd*S1/n + gamma*istd^2*S2/n*mean.
// Coefficients are zero-padded to Cpad.  eval-mode BN (training == 0): dx = gamma*istd*dy.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const double* __restrict__ partials, int R, int pstride, double count, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ istd, int C, int Cpad, int training,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ cA, float* __restrict__ cB,
    float* __restrict__ cC, float* __restrict__ sB, float* __restrict__ sC, int s_accumulate, int c_lo, int c_hi,
    const float* __restrict__ beta, const float* __restrict__ Wc, const float* __restrict__ dWc, int w_rows,
    int* __restrict__ cond /* [C] or NULL */, int* __restrict__ any_cond /* 1 int or NULL */) {
  // one wavefront per channel; lanes stride over the R partial rows; only channels [c_lo, c_hi) are touched
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = c_lo + blockIdx.x * 4 + wave;
  if (c >= Cpad || c >= c_hi) return;
  float a = 0.f, b = 0.f, d = 0.f;
  if (c < C) {
    double S1 = 0.0, S2 = 0.0;
    for (int g = lane; g < R; g += 64) {
      S1 += partials[(size_t)g * pstride + 2 * c];
      S2 += partials[(size_t)g * pstride + 2 * c + 1];
    }
    double Q = 0.0;   // Wc != NULL: sum_o W[o][c] * dW[o][c] = sum_p dy * bn(x) (see the header)
    if (Wc)
      for (int o = lane; o < w_rows; o += 64) Q += (double)Wc[(size_t)o * C + c] * (double)dWc[(size_t)o * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      S1 += __shfl_xor(S1, o, 64);
      S2 += __shfl_xor(S2, o, 64);
      Q += __shfl_xor(Q, o, 64);
    }
    if (Wc) {
      // |gamma| below 1e-12: bn(x) carries no xhat at f32 resolution and the quotient would only amplify rounding noise
      const double ga0 = gamma[c], bs = (double)beta[c] * S1, num = Q - bs;
      S2 = fabs(ga0) >= 1e-12 ? num / ga0 : 0.0;
      // Conditioning of the quotient.  Q = sum_p dy*(gamma*xhat + beta) and beta*S1 share the term beta * sum_p dy, which
      // cancels analytically but is evaluated twice in f32 with different summation orders: num carries a noise of about
      // |beta| * 1e-6 * sqrt(n) * rms(dy) against a signal gamma * S2 ~ |gamma| * sqrt(n) * rms(dy), so the relative error
      // of dgamma = S2 is ~1e-6 * |beta / gamma| (measured on the MI355X: 1 % at gamma = 1e-5, beta = .15; garbage at
      // 1e-7; and gamma = 0 has no quotient at all).  Note that |Q| + |beta*S1| is NOT a usable noise scale: with gamma ~ 0
      // and beta > 0 the ReLU passes every pixel and sum_p dy = W^T sum_p dz = 0 (BatchNorm backward), both terms are
      // noise themselves.  Channels with |gamma| < 1e-3 * |beta| (routine in trained DenseNets, never drawn by the other
      // tests) are FLAGGED and eml_dense_bn_dgamma_direct_f32 recomputes their S2 as the direct f64 sum of dy * xhat.  The
      // coefficients cB, cC below only need gamma * S2 = num, which is fine in absolute terms: they keep this value.
      if (cond && lane == 0) {
        const int ill = fabs(ga0) < 1e-3 * fabs((double)beta[c]) || ga0 == 0.0;
        cond[c] = ill;
        if (ill && any_cond) *any_cond = 1;   // benign race: every writer stores 1
      }
    }
    if (lane == 0) {
      dgamma[c] = (float)S2;
      dbeta[c] = (float)S1;
    }
    const double ga = gamma[c], is = istd[c], mu = mean[c];
    a = (float)(ga * is);
    if (training) {
      b = (float)(-ga * is * is * S2 / count);
      d = (float)(-ga * is * S1 / count + ga * is * is * S2 / count * mu);
    }
  }
  if (lane != 0) return;
  if (cA) {
    cA[c] = a;
    cB[c] = b;
    cC[c] = d;
  }
  if (sB) {  // deferred x-affine of the block gradient: Gfull = G + sB*x + sC
    sB[c] = (s_accumulate ? sB[c] : 0.f) + b;
    sC[c] = (s_accumulate ? sC[c] : 0.f) + d;
  }
}

// Sum R partial rows into a weight-gradient tensor (64 outputs x 4 row slices per block, f64).
// mode 0: dW1  src k*48+o (k<dim0=Cin, o<dim1=n_valid)     -> dW[(n0+o)*Cin + k]
// mode 1: dW2  src ((tap*3+mc)*16+ci)*16+o (o<12)            -> dW2[(o*48 + 16mc+ci)*9 + tap]
// mode 2: dW0  src t*32+o (t<27, o<dim1=C0)                  -> dW0[o*27 + t]
// mode 3: dW2, tap-packed rows: src ((t*3+n)*16+i)*16+j, row 16t+i = 12 tap + o (< 108), c = 16n+j -> dW2[(o*48 + c)*9 + tap]
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, int R, size_t row_stride,
                                                          int mode, int dim0, int dim1, int n0,
                                                          float* __restrict__ out) {
  __shared__ double red[4][64];
  const int tid = threadIdx.x, e = blockIdx.x * 64 + (tid & 63), slice = tid >> 6;
  size_t src = 0, dst = 0;
  bool valid = false;
  if (mode == 0) {
    const int k = e / 48, o = e - 48 * k;
    valid = k < dim0 && o < dim1;
    src = (size_t)e;
    dst = (size_t)(n0 + o) * dim0 + k;
  } else if (mode == 1) {
    const int o = e & 15, ci = (e >> 4) & 15, pair = e >> 8, tap = pair / 3, mc = pair - 3 * tap;
    valid = pair < 27 && o < 12;
    src = (size_t)e;
    dst = (size_t)(o * 48 + 16 * mc + ci) * 9 + tap;
  } else if (mode == 3) {
    const int j = e & 15, i = (e >> 4) & 15, tn = e >> 8, t = tn / 3, n = tn - 3 * t, row = 16 * t + i;
    valid = tn < 21 && row < 108;
    src = (size_t)e;
    dst = (size_t)((row % 12) * 48 + 16 * n + j) * 9 + row / 12;
  } else {
    const int t = e >> 5, o = e & 31;
    valid = t < 27 && o < dim1;
    src = (size_t)e;
    dst = (size_t)o * 27 + t;
  }
  double s = 0.0;
  if (valid) {
#pragma unroll 8
    for (int r = slice; r < R; r += 4) s += (double)partial[(size_t)r * row_stride + src];
  }
  red[slice][tid & 63] = s;
  __syncthreads();
  if (slice == 0 && valid) out[dst] = (float)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
}

// =============================================================================== conv1x1 backward: weight
// dW[k][o] = sum_p a[p][k] * dz[p][o];  a = relu(s1*x + t1) (POOL: 2x2 mean of it),
// dz[p][o] = cA[o]*DY[p][o] + cB[o]*Zr[p][o] + cC[o].  D[i][j=o], MFMA-k = pixel.
//   * dz of a 64-pixel chunk is built once per workgroup in LDS (double-buffered, one barrier per
//     chunk) and read as B fragments by all four waves;
//   * the A operand comes straight from global memory as float2: MFMA's row index is only a label,
//     so lane i of a 32-channel group loads channels (2i, 2i+1) of its pixel -- 128 contiguous
//     bytes per pixel -- and feeds them to two accumulator tiles whose row i means channel 2i+t;
//   * the Kp/32 channel groups are dealt round-robin to the 4 waves (<= 3 each).
//   * NARROW (dense layers, the upper layer of a backward pair): the dz tile sitting in LDS also feeds the NARROW data
//     pass -- the finished gradient of the lower layer's 12 output channels [k_lo, k_lo + 12),
//     N12[p][c] = G[p][k_lo+c] + scale1*mask*(dz[p] . W1[:, k_lo+c]) -- as 12 extra MFMAs per wave and chunk (wave w
//     owns pixel tile w): what was a separate streaming kernel re-reading dz (48 floats per pixel) and a 12-channel
//     slice of x is an epilogue on data this kernel already holds (the x slice is the tail of the row it just read).
struct NarrowArgs {
  const float* W1;     // [48][Cin] (PyTorch layout)
  const float* G;      // block gradient, read only
  float* N12;          // (P, 12) out
  double* partials;    // [grid][Kp][2]: S1 of channels k_lo .. k_lo+11 (S2 slot = 0: it comes from the weight gradient)
  int Cin, k_lo, ldg;
};
// VEC (round 4): floats per lane of the x operand.  2 = round 1's float2 (lane i of a 32-channel group
// carries channels 2i, 2i+1: a wave instruction reads 4 rows x 128 B); 4 = float4 over 64-channel groups (4 rows x 256 B):
// half as many vector-memory instructions for the operand that is 80 % of this kernel's bytes.  The access-shape probe
// (profiles/r04_row_access_probe_wgrad.txt) streams the float2 shape at 4.6-5.0 TB/s and the float4 shape at 5.5-5.9 at
// k = 208 / 224 (2.7-2.95 against 3.4-3.65 with two workgroups per CU) -- but the kernel itself did not get faster with it
// (see the launcher): kept as an A/B build (EML_W1_VEC=4), the float2 form runs.
template <bool POOL, bool NARROW = false, int VEC = 2>
__global__ __launch_bounds__(256, (NARROW || VEC == 4) ? 2 : 1) void conv1x1_bwd_weight_kernel(
    const float* __restrict__ X, int ldx, int P, int Hin, int Win, int Kp, const float* __restrict__ scale1,
    const float* __restrict__ shift1, const float* __restrict__ DY, int ld_dy, const float* __restrict__ Zr,
    int ld_z, const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC, int n_valid,
    int n_load /* columns (multiple of 4) that may be read without leaving the DY / Zr rows */,
    float* __restrict__ partial /*[grid][Kp][48]*/,
    float* __restrict__ dz_out /* NULL, or (P,48): the rebuilt dz is materialised here for the data-gradient passes
                                  (may alias DY: every element is read and written by the same thread) */,
    NarrowArgs na) {
  __shared__ __attribute__((aligned(16))) float dz_l[2][64 * 48];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  static_assert(VEC == 2 || (VEC == 4 && !POOL), "float4 operand: dense layers only");
  constexpr int GW = 16 * VEC;            // channels per group
  constexpr int MAXG = VEC == 2 ? 3 : 2;  // groups per wave (Kp <= 384)
  const int ngroups = (Kp + GW - 1) / GW;
  // narrow pass: A fragments (rows = the 12 channels, k = o), BN1 affine of this lane's 4 channels, f64 statistics
  float nsk[4] = {0.f, 0.f, 0.f, 0.f}, ntk[4] = {0.f, 0.f, 0.f, 0.f};
  double ns1[4] = {0.0, 0.0, 0.0, 0.0};
  const int kq = min(kk, 2);   // lanes kk = 3 hold MFMA rows 12..15: padding (clamped addresses, masked results)
  if constexpr (NARROW) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      nsk[g] = kk < 3 ? scale1[na.k_lo + 4 * kq + g] : 0.f;
      ntk[g] = kk < 3 ? shift1[na.k_lo + 4 * kq + g] : 0.f;
    }
  }

  float sv[MAXG][VEC], tv[MAXG][VEC];
  int coff[MAXG];
  bool gv[MAXG];
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int cg = wave + 4 * i;
    const int ch = GW * cg + VEC * r;
    gv[i] = cg < ngroups && ch < Kp;
    coff[i] = gv[i] ? ch : 0;
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
      sv[i][t] = gv[i] ? scale1[ch + t] : 0.f;
      tv[i][t] = gv[i] ? shift1[ch + t] : 0.f;
    }
  }
  // dz staging role: pixel = tid >> 2, columns 4*(q + 4j), j = 0..2.  NARROW: the BN2-backward affine (cA, cB, cC; 0
  // past n_valid) and the narrow pass's weight fragments live in LDS (read once per chunk): the plain kernel sits at the
  // 256-register line of two waves per SIMD (180 VGPRs + 72 AGPRs) and the narrow epilogue needs ~40 more.  The plain
  // variant keeps them in registers -- moved to LDS too it lost 2 ms/step (measured).
  const int spix = tid >> 2, sq = tid & 3;
  __shared__ __attribute__((aligned(16))) float co_l[3 * 48];
  __shared__ float wn_l[NARROW ? 12 * 64 : 1];
  constexpr bool CO_LDS = NARROW || VEC == 4;   // the float4 variant needs the registers for its accumulators
  float4 sa_r[CO_LDS ? 1 : 3], sb_r[CO_LDS ? 1 : 3], sc_r[CO_LDS ? 1 : 3];   // plain variant: the affine in registers
  if constexpr (CO_LDS) {
    if (tid < 48) {
      const bool v = tid < n_valid;
      co_l[tid] = v ? cA[tid] : 0.f;
      co_l[48 + tid] = v ? cB[tid] : 0.f;
      co_l[96 + tid] = v ? cC[tid] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 4 * (sq + 4 * j);
      float a[4], b[4], c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool v = col + e < n_valid;
        a[e] = v ? cA[col + e] : 0.f;
        b[e] = v ? cB[col + e] : 0.f;
        c[e] = v ? cC[col + e] : 0.f;
      }
      sa_r[j] = make_float4(a[0], a[1], a[2], a[3]);
      sb_r[j] = make_float4(b[0], b[1], b[2], b[3]);
      sc_r[j] = make_float4(c[0], c[1], c[2], c[3]);
    }
  }
  if constexpr (NARROW) {   // A fragments (rows = the 12 channels, k = o): entry (st, kk, r) = W1[4*st + kk][k_lo + r]
    for (int e = tid; e < 12 * 64; e += 256) {
      const int st = e >> 6, l = e & 63, rr = l & 15, k4 = l >> 4;
      wn_l[e] = rr < 12 ? na.W1[(size_t)(4 * st + k4) * na.Cin + na.k_lo + rr] : 0.f;
    }
  }
  __syncthreads();
  f32x4 acc[MAXG][VEC][3];
#pragma unroll
  for (int i = 0; i < MAXG; ++i)
#pragma unroll
    for (int t = 0; t < VEC; ++t)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[i][t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int Wo = Win >> 1, Ho = Hin >> 1;
  const int nchunks = (P + 63) >> 6;
  const int ngw = (ngroups > wave) ? (ngroups - wave + 3) >> 2 : 0;  // channel groups of this wave (uniform)

  // dz tile of a chunk: global -> registers (issued early) -> affine -> LDS (written late)
  float4 rdy[3], rzr[3];
  bool rpv = false;
  int rp = 0;
  auto stage_load = [&](int chunk) {
    const int p = chunk * 64 + spix;
    rpv = chunk < nchunks && p < P;
    const size_t pc = rpv ? p : 0;
    rp = (int)pc;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 4 * (sq + 4 * j);
      const bool cv = col < n_load;
      rdy[j] = cv ? *reinterpret_cast<const float4*>(DY + pc * ld_dy + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      rzr[j] = cv ? *reinterpret_cast<const float4*>(Zr + pc * ld_z + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_write = [&](float* dzb) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 4 * (sq + 4 * j);
      float4 sa, sb, sc;
      if constexpr (CO_LDS) {
        sa = *reinterpret_cast<const float4*>(co_l + col);
        sb = *reinterpret_cast<const float4*>(co_l + 48 + col);
        sc = *reinterpret_cast<const float4*>(co_l + 96 + col);
      } else {
        sa = sa_r[j];
        sb = sb_r[j];
        sc = sc_r[j];
      }
      float4 v;
      v.x = rpv ? fmaf(sa.x, rdy[j].x, fmaf(sb.x, rzr[j].x, sc.x)) : 0.f;
      v.y = rpv ? fmaf(sa.y, rdy[j].y, fmaf(sb.y, rzr[j].y, sc.y)) : 0.f;
      v.z = rpv ? fmaf(sa.z, rdy[j].z, fmaf(sb.z, rzr[j].z, sc.z)) : 0.f;
      v.w = rpv ? fmaf(sa.w, rdy[j].w, fmaf(sb.w, rzr[j].w, sc.w)) : 0.f;
      *reinterpret_cast<float4*>(dzb + spix * 48 + col) = v;
      if (dz_out && rpv) *reinterpret_cast<float4*>(dz_out + (size_t)rp * 48 + col) = v;
    }
  };
  // The chunk loop, specialised on the (wave-uniform) number of channel groups so that the operand
  // loads of UQ pixel-quads are issued back to back with no control flow between them.
  auto chunk_loop = [&](auto ngw_c) {
    constexpr int NGW = decltype(ngw_c)::value;
#ifdef EML_WGRAD_DEEP   // experiment build: half-size batches, four buffers, three batches (24 pixels) in flight
    constexpr int UQ = POOL ? 1 : 2;
    constexpr int NBUF = POOL ? 2 : 4;
#else
    constexpr int UQ = POOL ? 1 : (VEC == 4 ? 2 : 4);   // pixel quads per batch (the same bytes in flight for both VEC)
    constexpr int NBUF = 2;
#endif
    constexpr int NS = POOL ? 4 : 1;      // input pixels per output pixel
    constexpr int NB = 16 / UQ;           // operand batches per 64-pixel chunk (a multiple of NBUF)
    constexpr int AHEAD = NBUF - 1;       // batches requested ahead of their MFMAs
    // x operand of one batch: UNCONDITIONAL loads from clamped pixels (validity is applied when the value is
    // used); they are requested AHEAD of their MFMAs, across chunk boundaries -- issued right before
    // use they exposed an HBM round trip per batch (ISA: global_load; s_waitcnt vmcnt; v_mfma).
    using xv_t = std::conditional_t<VEC == 2, float2, float4>;
    xv_t xr[NBUF][UQ][NGW > 0 ? NGW : 1][NS];
    auto load_batch = [&](int chunk, int q0, xv_t (&dst)[UQ][NGW > 0 ? NGW : 1][NS]) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        const int pc = min(chunk * 64 + 4 * (q0 + u) + kk, P - 1);
        const float* xp;
        if constexpr (POOL) {
          const int b = pc / (Ho * Wo), rem = pc - b * (Ho * Wo);
          const int oy = rem / Wo, ox = rem - oy * Wo;
          xp = X + ((size_t)(b * Hin + 2 * oy) * Win + 2 * ox) * ldx;
        } else {
          xp = X + (size_t)pc * ldx;
        }
#pragma unroll
        for (int i = 0; i < NGW; ++i)
#pragma unroll
          for (int sub = 0; sub < NS; ++sub)
            dst[u][i][sub] = *reinterpret_cast<const xv_t*>(xp + ((sub >> 1) * (size_t)Win + (sub & 1)) * ldx + coff[i]);
      }
    };
    stage_load(blockIdx.x);
    stage_write(dz_l[0]);
    if constexpr (NGW > 0) {
#pragma unroll
      for (int a = 0; a < AHEAD; ++a) load_batch(min((int)blockIdx.x, nchunks - 1), a * UQ, xr[a]);
    }
    __syncthreads();
    int it = 0;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x, ++it) {
      const float* dzb = dz_l[it & 1];
      stage_load(chunk + gridDim.x);  // next chunk's loads fly during this chunk's MFMAs
      // narrow pass operands of this wave's pixel tile (x slice: the tail of rows this kernel reads anyway; G slice):
      // requested now, used after the chunk's MFMAs
      float2 nx[2], ng[2];
      const int pn = chunk * 64 + 16 * wave + r;
      if constexpr (NARROW) {
        const size_t pc = (size_t)min(pn, P - 1);
        const float* xs = X + pc * ldx + na.k_lo + 4 * kq;
        // ldg == 12: G is a compact (P, 12) tensor of exactly these channels (the data-gradient pass's `top` output)
        const float* gs = na.G + pc * na.ldg + (na.ldg == 12 ? 0 : na.k_lo) + 4 * kq;
        nx[0] = *reinterpret_cast<const float2*>(xs);
        nx[1] = *reinterpret_cast<const float2*>(xs + 2);
        ng[0] = *reinterpret_cast<const float2*>(gs);
        ng[1] = *reinterpret_cast<const float2*>(gs + 2);
      }
      if constexpr (NGW > 0) {
        float bzq[2][3];
#pragma unroll
        for (int n = 0; n < 3; ++n) bzq[0][n] = dzb[kk * 48 + 16 * n + r];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
          const int q0 = bi * UQ;
          if (bi + AHEAD < NB)
            load_batch(chunk, q0 + AHEAD * UQ, xr[(bi + AHEAD) % NBUF]);
          else
            load_batch(min(chunk + (int)gridDim.x, nchunks - 1), (bi + AHEAD - NB) * UQ, xr[(bi + AHEAD) % NBUF]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
            const int pl = 4 * (q0 + u) + kk;
            const bool pvu = chunk * 64 + pl < P;
            // the dz fragments of the NEXT pixel quad are requested before this quad's MFMAs (round 4; quad 0 of a chunk
            // is read right after the chunk's barrier, below)
            float bz[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) bz[n] = bzq[(q0 + u) & 1][n];
            if (q0 + u + 1 < 16) {
#pragma unroll
              for (int n = 0; n < 3; ++n) bzq[(q0 + u + 1) & 1][n] = dzb[(pl + 4) * 48 + 16 * n + r];
            }
#pragma unroll
            for (int i = 0; i < NGW; ++i) {
              float av[VEC];
#pragma unroll
              for (int t = 0; t < VEC; ++t) av[t] = 0.f;
#pragma unroll
              for (int sub = 0; sub < NS; ++sub) {
                const xv_t xv = xr[bi % NBUF][u][i][sub];
                av[0] += fmaxf(fmaf(xv.x, sv[i][0], tv[i][0]), 0.f);
                av[1] += fmaxf(fmaf(xv.y, sv[i][1], tv[i][1]), 0.f);
                if constexpr (VEC == 4) {
                  av[2] += fmaxf(fmaf(xv.z, sv[i][2], tv[i][2]), 0.f);
                  av[3] += fmaxf(fmaf(xv.w, sv[i][3], tv[i][3]), 0.f);
                }
              }
#pragma unroll
              for (int t = 0; t < VEC; ++t) {
                if constexpr (POOL) av[t] *= 0.25f;
                if (!pvu) av[t] = 0.f;
              }
#pragma unroll
              for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int t = 0; t < VEC; ++t) acc[i][t][n] = mfma16(av[t], bz[n], acc[i][t][n]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (NARROW) {
        // an unmasked use of the narrow operands HERE: their only other use feeds the exec-masked N12 store, and the compiler
        // sank the G-slice load from the top of the chunk into that masked block -- issued after the chunk's MFMAs and waited
        // for at once (s_waitcnt vmcnt(0): behind every x request in flight as well), an HBM round trip per 64-pixel chunk
        asm volatile("" ::"v"(ng[0].x), "v"(ng[0].y), "v"(ng[1].x), "v"(ng[1].y), "v"(nx[0].x), "v"(nx[0].y), "v"(nx[1].x), "v"(nx[1].y));
        f32x4 an = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 12; ++st) an = mfma16(wn_l[st * 64 + lane], dzb[(16 * wave + r) * 48 + 4 * st + kk], an);
        const bool nv = pn < P && kk < 3;
        const float xg[4] = {nx[0].x, nx[0].y, nx[1].x, nx[1].y}, gg[4] = {ng[0].x, ng[0].y, ng[1].x, ng[1].y};
        float o4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float dam = (nv && fmaf(xg[g], nsk[g], ntk[g]) > 0.f) ? an[g] : 0.f;
          o4[g] = fmaf(nsk[g], dam, gg[g]);
          ns1[g] += (double)dam;
        }
        // (32-bit offset off the uniform base, P * 12 < 2^31: the 64-bit per-lane pointer was spilled -- a reload + vmcnt(0) here)
        if (nv) *reinterpret_cast<float4*>(na.N12 + ((unsigned)pn * 12u + 4u * (unsigned)kk)) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
      stage_write(dz_l[(it + 1) & 1]);
      // LDS-only barrier: __syncthreads() would also wait (vmcnt(0)) for the x operands of the NEXT chunk's first
      // batch, requested a moment ago -- one exposed HBM round trip per 64-pixel chunk
      eml::lds_barrier();
    }
  };
  switch (ngw) {  // wave-uniform
    case 0: chunk_loop(std::integral_constant<int, 0>{}); break;
    case 1: chunk_loop(std::integral_constant<int, 1>{}); break;
    case 2: chunk_loop(std::integral_constant<int, 2>{}); break;
    default: chunk_loop(std::integral_constant<int, MAXG>{}); break;
  }
  float* out = partial + (size_t)blockIdx.x * Kp * 48;
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int cg = wave + 4 * i;
#pragma unroll
    for (int t = 0; t < VEC; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = GW * cg + VEC * (4 * kk + g) + t;  // D row 4kk+g of tile t
        if (cg < ngroups && ch < Kp) {
#pragma unroll
          for (int n = 0; n < 3; ++n) out[(size_t)ch * 48 + 16 * n + r] = acc[i][t][n][g];
        }
      }
  }
  if constexpr (NARROW) {   // S1 of the 12 narrow channels: over the 16 pixel lanes, then the 4 waves
    __syncthreads();        // everyone is done with dz_l
    double* red = reinterpret_cast<double*>(&dz_l[0][0]);   // [4 waves][12]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ns1[g] += shfl_xor_d(ns1[g], o);
      if (r == 0 && kk < 3) red[wave * 12 + 4 * kk + g] = ns1[g];
    }
    __syncthreads();
    if (tid < 12) {
      double* dst = na.partials + ((size_t)blockIdx.x * Kp + na.k_lo + tid) * 2;
      dst[0] = (red[tid] + red[12 + tid]) + (red[24 + tid] + red[36 + tid]);
      dst[1] = 0.0;
    }
  }
}

// =============================================================================== conv1x1 backward: data
// da[p][k] = sum_o dz[p][o] W[o][k], masked by relu(bn1(x)) > 0, and -- since the dy-coefficient of
// a BatchNorm backward, gamma*istd == scale1, needs no statistics -- accumulated straight into the
// block gradient:  G[p][k] (+)= scale1[k] * dam[p][k].  The statistics-dependent part of BN1's
// backward (cB*x + cC) is affine in x with per-channel coefficients, so it is only summed into
// sB/sC by bn_bwd_finalize and applied once per channel by grad_materialize.
// Partial sums S1 = sum dam, S2 = sum dam * xhat per input channel (xhat = (x-mean)*istd).
// The MFMA computes D^T (rows = channels, cols = pixels) so a lane owns 4 CONSECUTIVE channels of
// one pixel: X / G are touched with 16-B accesses.
// Wd: weights in fragment order [Kp/16][Ko/16][4][16][4] = W[o=16jo+4kk+t][k=16nt+col].
template <bool POOL, bool RES /* Ko == 48: dz fragments stay in registers across channel chunks */, int NCH = 4>
__global__ __launch_bounds__(256) void conv1x1_bwd_data_kernel(
    const float* __restrict__ DY, int ld_dy, const float* __restrict__ Zr, int ld_z, const float* __restrict__ cA,
    const float* __restrict__ cB, const float* __restrict__ cC, int Ko, const float* __restrict__ Wd,
    const float* __restrict__ X, int ldx, const float* __restrict__ scale1, const float* __restrict__ shift1,
    const float* __restrict__ mean, const float* __restrict__ istd, int P, int Hin, int Win, int Kp,
    float* __restrict__ Gd, int ldg, int accumulate, double* __restrict__ partials /*[grid][Kp][2]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* sacc = reinterpret_cast<double*>(smem);  // [4 waves][Kp][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  for (int e = tid; e < 4 * Kp * 2; e += 256) sacc[e] = 0.0;
  __syncthreads();
  double* my = sacc + (size_t)wave * Kp * 2;
  const int nnt = Kp >> 4, njo = Ko >> 4;
  const int ntiles = (P + 255) >> 8;
  const int Wo = Win >> 1, Ho = Hin >> 1;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p0 = tile * 256 + wave * 64;
    long prow[4];
    size_t pin[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      prow[m] = min(p0 + 16 * m + r, P - 1);
      if constexpr (POOL) {
        const int b = (int)(prow[m] / (Ho * Wo)), rem = (int)(prow[m] - (long)b * (Ho * Wo));
        const int oy = rem / Wo, ox = rem - oy * Wo;
        pin[m] = (size_t)(b * Hin + 2 * oy) * Win + 2 * ox;
      } else {
        pin[m] = (size_t)prow[m];
      }
    }
    auto load_dz = [&](int jo, float4 (&dz)[4]) {
      const int ch = 16 * jo + 4 * kk;
      const float4 a4 = *reinterpret_cast<const float4*>(cA + ch);
      const float4 b4 = *reinterpret_cast<const float4*>(cB + ch);
      const float4 c4 = *reinterpret_cast<const float4*>(cC + ch);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float4 dy = *reinterpret_cast<const float4*>(DY + prow[m] * ld_dy + ch);
        const float4 zr = *reinterpret_cast<const float4*>(Zr + prow[m] * ld_z + ch);
        dz[m].x = fmaf(a4.x, dy.x, fmaf(b4.x, zr.x, c4.x));
        dz[m].y = fmaf(a4.y, dy.y, fmaf(b4.y, zr.y, c4.y));
        dz[m].z = fmaf(a4.z, dy.z, fmaf(b4.z, zr.z, c4.z));
        dz[m].w = fmaf(a4.w, dy.w, fmaf(b4.w, zr.w, c4.w));
      }
    };
    float4 dzr[3][4];
    if constexpr (RES) {
#pragma unroll
      for (int jo = 0; jo < 3; ++jo) load_dz(jo, dzr[jo]);
    }
    for (int nt0 = 0; nt0 < nnt; nt0 += NCH) {
      f32x4 acc[4][NCH];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NCH; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto mma = [&](int jo, const float4 (&dz)[4]) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
          const int nt = min(nt0 + n, nnt - 1);
          const float4 w = *reinterpret_cast<const float4*>(Wd + ((((size_t)nt * njo + jo) * 4 + kk) * 16 + r) * 4);
#pragma unroll
          for (int t = 0; t < 4; ++t)  // A = W^T fragment, B = dz fragment  ->  D[channel][pixel]
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][n] = mfma16(f4c(w, t), f4c(dz[m], t), acc[m][n]);
        }
      };
      if constexpr (RES) {
#pragma unroll
        for (int jo = 0; jo < 3; ++jo) mma(jo, dzr[jo]);
      } else {
        for (int jo = 0; jo < njo; ++jo) {
          float4 dz[4];
          load_dz(jo, dz);
          mma(jo, dz);
        }
      }
      // epilogue: this lane owns channels k4..k4+3 (k4 = 16nt + 4kk) of pixel p0 + 16m + r
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        const int nt = nt0 + n;
        if (nt < nnt) {
          const int k4 = 16 * nt + 4 * kk;
          const float4 sk = *reinterpret_cast<const float4*>(scale1 + k4);
          const float4 tk = *reinterpret_cast<const float4*>(shift1 + k4);
          const float4 mu = *reinterpret_cast<const float4*>(mean + k4);
          const float4 is = *reinterpret_cast<const float4*>(istd + k4);
          float l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (p0 + 16 * m + r < P) {
              const float da[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
              constexpr int NSUB = POOL ? 4 : 1;
#pragma unroll
              for (int sub = 0; sub < NSUB; ++sub) {
                const size_t pi = pin[m] + (sub >> 1) * (size_t)Win + (sub & 1);
                const float4 xv = *reinterpret_cast<const float4*>(X + pi * ldx + k4);
                float4* gp = reinterpret_cast<float4*>(Gd + pi * ldg + k4);
                float4 g = accumulate ? *gp : make_float4(0.f, 0.f, 0.f, 0.f);
                const float f = POOL ? 0.25f : 1.0f;
                const float d0 = (fmaf(xv.x, sk.x, tk.x) > 0.f) ? f * da[0] : 0.f;
                const float d1 = (fmaf(xv.y, sk.y, tk.y) > 0.f) ? f * da[1] : 0.f;
                const float d2 = (fmaf(xv.z, sk.z, tk.z) > 0.f) ? f * da[2] : 0.f;
                const float d3 = (fmaf(xv.w, sk.w, tk.w) > 0.f) ? f * da[3] : 0.f;
                g.x = fmaf(sk.x, d0, g.x);
                g.y = fmaf(sk.y, d1, g.y);
                g.z = fmaf(sk.z, d2, g.z);
                g.w = fmaf(sk.w, d3, g.w);
                *gp = g;
                l1[0] += d0;
                l1[1] += d1;
                l1[2] += d2;
                l1[3] += d3;
                l2[0] = fmaf(d0, (xv.x - mu.x) * is.x, l2[0]);
                l2[1] = fmaf(d1, (xv.y - mu.y) * is.y, l2[1]);
                l2[2] = fmaf(d2, (xv.z - mu.z) * is.z, l2[2]);
                l2[3] = fmaf(d3, (xv.w - mu.w) * is.w, l2[3]);
              }
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            l1[g] = eml::row16_sum(l1[g]);
            l2[g] = eml::row16_sum(l2[g]);
            if (r == 0) {
              my[2 * (k4 + g)] += (double)l1[g];
              my[2 * (k4 + g) + 1] += (double)l2[g];
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < Kp * 2; e += 256)
    partials[(size_t)blockIdx.x * Kp * 2 + e] =
        (sacc[e] + sacc[(size_t)Kp * 2 + e]) + (sacc[(size_t)2 * Kp * 2 + e] + sacc[(size_t)3 * Kp * 2 + e]);
}

// ------------------------------------------------------------------------------- transition (pooled) dgrad
// The general kernel above runs the transitions (Ko = 112..176 output channels, 2x2 un-pooling) at one wave per SIMD
// (296 VGPRs) and loads x inside the exec-masked epilogue.  This one is built like the dense-layer pass: 32 pooled
// pixels per wave, two 16-channel tiles per chunk, the chunk's 16 x quads (4 input pixels per pooled pixel) requested
// before its MFMAs, the next output-channel group's dz / weight fragments requested one step ahead, per-channel
// vectors in LDS, unconditional operand use, DPP row reductions -- 2 waves per SIMD and no exposed round trips.
// MASKED: the ReLU mask comes from the 16-bit words pool_act_kernel stored (bit 4*sub + g of word [pooled pixel][channel
// quad]); x, mean, istd are not read and only S1 is accumulated (S2 from the transition conv's weight gradient).
// NJO > 0 (= Ko / 16, round 3): the dz fragments of the tile's 32 pooled pixels (dz = cA*dY + cB*T + cC, Ko channels) are
// built ONCE per tile and stay in registers across the channel chunks, like the dense-layer pass does -- the streaming
// form (NJO = 0) re-read the dY and T rows and re-applied the affine for every one of the Kp/32 chunks (7 times per
// tile at transition 1: 6 KB of L2 -> L1 traffic per pooled pixel against 3.5 KB of output).
template <int MT, int NCH, bool ACC /* G += instead of G = */, bool MASKED = false, int NJO = 0>
__global__ __launch_bounds__(256, 2) void transition_bwd_data_kernel(
    const float* __restrict__ DY, int ld_dy, const float* __restrict__ Zr, int ld_z, const float* __restrict__ cA,
    const float* __restrict__ cB, const float* __restrict__ cC, int Ko, const float* __restrict__ Wd,
    const float* __restrict__ X, int ldx, const float* __restrict__ scale1, const float* __restrict__ shift1,
    const float* __restrict__ mean, const float* __restrict__ istd, int P, int Hin, int Win, int Kp,
    float* __restrict__ Gd, int ldg, double* __restrict__ partials /*[grid][Kp][2]*/,
    const unsigned short* __restrict__ mask16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* sacc = reinterpret_cast<double*>(smem);                  // [4 waves][Kp][2]
  float* vec_l = reinterpret_cast<float*>(sacc + (size_t)4 * Kp * 2);  // scale1, shift1, mean, istd [Kp]; cA, cB, cC [Ko]
  float* co_l = vec_l + 4 * Kp;
  float* stg = co_l + 3 * Ko + (size_t)(threadIdx.x >> 6) * 16 * 36;   // [4 waves][16][36]: MASKED && !ACC only
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  for (int e = tid; e < 4 * Kp * 2; e += 256) sacc[e] = 0.0;
  for (int e = tid; e < Kp; e += 256) {
    vec_l[e] = scale1[e];
    vec_l[Kp + e] = shift1[e];
    vec_l[2 * Kp + e] = MASKED ? 0.f : mean[e];
    vec_l[3 * Kp + e] = MASKED ? 0.f : istd[e];
  }
  for (int e = tid; e < Ko; e += 256) {
    co_l[e] = cA[e];
    co_l[Ko + e] = cB[e];
    co_l[2 * Ko + e] = cC[e];
  }
  __syncthreads();
  double* my = sacc + (size_t)wave * Kp * 2;
  const int nnt = Kp >> 4, njo = Ko >> 4;
  constexpr int TP = 64 * MT;
  const int ntiles = (P + TP - 1) / TP;
  const int Wo = Win >> 1, Ho = Hin >> 1;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p0 = tile * TP + wave * 16 * MT;
    long prow[MT];
    size_t pin[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      pv[m] = p0 + 16 * m + r < P;
      prow[m] = min(p0 + 16 * m + r, P - 1);
      const int b = (int)(prow[m] / (Ho * Wo)), rem = (int)(prow[m] - (long)b * (Ho * Wo));
      const int oy = rem / Wo, ox = rem - oy * Wo;
      pin[m] = (size_t)(b * Hin + 2 * oy) * Win + 2 * ox;   // top-left input pixel of the pooling window
    }
    // staged stores (MASKED && !ACC): the rows lane l writes are those of lanes l >> 3 and (l >> 3) + 8
    size_t pinx[MT][2];
    bool pvx[MT][2];
    if constexpr (MASKED && !ACC) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int src = (lane >> 3) + 8 * h;
          pinx[m][h] = (size_t)__shfl((unsigned long long)pin[m], src, 64);
          pvx[m][h] = __shfl((int)pv[m], src, 64) != 0;
        }
    }
    float4 dzr[NJO > 0 ? NJO : 1][MT];
    if constexpr (NJO > 0) {
#pragma unroll
      for (int jo = 0; jo < NJO; ++jo) {
        const int ch = 16 * jo + 4 * kk;
        const float4 a4 = *reinterpret_cast<const float4*>(co_l + ch);
        const float4 b4 = *reinterpret_cast<const float4*>(co_l + Ko + ch);
        const float4 c4 = *reinterpret_cast<const float4*>(co_l + 2 * Ko + ch);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4 y4 = *reinterpret_cast<const float4*>(DY + prow[m] * ld_dy + ch);
          const float4 z4 = *reinterpret_cast<const float4*>(Zr + prow[m] * ld_z + ch);
          dzr[jo][m].x = fmaf(a4.x, y4.x, fmaf(b4.x, z4.x, c4.x));
          dzr[jo][m].y = fmaf(a4.y, y4.y, fmaf(b4.y, z4.y, c4.y));
          dzr[jo][m].z = fmaf(a4.z, y4.z, fmaf(b4.z, z4.z, c4.z));
          dzr[jo][m].w = fmaf(a4.w, y4.w, fmaf(b4.w, z4.w, c4.w));
        }
      }
    }
    for (int nt0 = 0; nt0 < nnt; nt0 += NCH) {
      // the chunk's x quads (and old G when accumulating): requested now, consumed after the MFMAs
      float4 xv[MASKED ? 1 : MT][MASKED ? 1 : NCH][4], gs[ACC ? MT : 1][ACC ? NCH : 1][4];
      unsigned mk[MT][NCH];
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        const int k4 = 16 * min(nt0 + n, nnt - 1) + 4 * kk;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr (MASKED) mk[m][n] = mask16[(size_t)prow[m] * (Kp >> 2) + (k4 >> 2)];
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            const size_t pi = pin[m] + (sub >> 1) * (size_t)Win + (sub & 1);
            if constexpr (!MASKED) xv[m][n][sub] = *reinterpret_cast<const float4*>(X + pi * ldx + k4);
            if constexpr (ACC) gs[m][n][sub] = *reinterpret_cast<const float4*>(Gd + pi * ldg + k4);
          }
        }
      }
      f32x4 acc[MT][NCH];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NCH; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      // output-channel groups, fragments of group jo+1 in flight during the MFMAs of group jo
      float4 ry[MT], rz[MT], wq[NCH];
      auto load_group = [&](int jo) {
        const int ch = 16 * jo + 4 * kk;
        if constexpr (NJO == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            ry[m] = *reinterpret_cast<const float4*>(DY + prow[m] * ld_dy + ch);
            rz[m] = *reinterpret_cast<const float4*>(Zr + prow[m] * ld_z + ch);
          }
        }
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
          const int nt = min(nt0 + n, nnt - 1);
          wq[n] = *reinterpret_cast<const float4*>(Wd + ((((size_t)nt * njo + jo) * 4 + kk) * 16 + r) * 4);
        }
      };
      load_group(0);
      auto group_step = [&](int jo, const float4 (&dz)[MT]) {
        float4 wc[NCH];
#pragma unroll
        for (int n = 0; n < NCH; ++n) wc[n] = wq[n];
        load_group(min(jo + 1, njo - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int n = 0; n < NCH; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][n] = mfma16(f4c(wc[n], t), f4c(dz[m], t), acc[m][n]);
      };
      if constexpr (NJO > 0) {
#pragma unroll
        for (int jo = 0; jo < NJO; ++jo) group_step(jo, dzr[jo]);
      } else {
        for (int jo = 0; jo < njo; ++jo) {
          const int ch = 16 * jo + 4 * kk;
          const float4 a4 = *reinterpret_cast<const float4*>(co_l + ch);
          const float4 b4 = *reinterpret_cast<const float4*>(co_l + Ko + ch);
          const float4 c4 = *reinterpret_cast<const float4*>(co_l + 2 * Ko + ch);
          float4 dz[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            dz[m].x = fmaf(a4.x, ry[m].x, fmaf(b4.x, rz[m].x, c4.x));
            dz[m].y = fmaf(a4.y, ry[m].y, fmaf(b4.y, rz[m].y, c4.y));
            dz[m].z = fmaf(a4.z, ry[m].z, fmaf(b4.z, rz[m].z, c4.z));
            dz[m].w = fmaf(a4.w, ry[m].w, fmaf(b4.w, rz[m].w, c4.w));
          }
          group_step(jo, dz);
        }
      }
      if constexpr (MASKED && !ACC) {
        // (round 6) G is write-only here: 16 rows x 64 bytes per store instruction are HALF lines into rows nobody holds in
        // L2 -- 2.15 TB/s on this shape against 3.9 for whole lines (profiles/r06_write_pattern.txt).  The two column tiles of
        // a (pooled pixel group, sub-pixel) go through a wave-private LDS tile (16 rows x 32 channels, stride 36) and leave as
        // two stores of 8 rows x one whole 128-byte line each: lane l -> row l >> 3 (+ 8), 16-byte piece l & 7.
        float l1n[NCH][4];
        float4 skn[NCH];
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
          skn[n] = *reinterpret_cast<const float4*>(vec_l + 16 * min(nt0 + n, nnt - 1) + 4 * kk);
#pragma unroll
          for (int g = 0; g < 4; ++g) l1n[n][g] = 0.f;
        }
        const int piece = lane & 7;
        const bool col_ok = nt0 + (piece >> 2) < nnt;             // the lane's column tile exists
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
            for (int n = 0; n < NCH; ++n) {
              const unsigned b = pv[m] ? mk[m][n] >> (4 * sub) : 0u;
              const float d0 = (b & 1u) ? 0.25f * acc[m][n][0] : 0.f, d1 = (b & 2u) ? 0.25f * acc[m][n][1] : 0.f;
              const float d2 = (b & 4u) ? 0.25f * acc[m][n][2] : 0.f, d3 = (b & 8u) ? 0.25f * acc[m][n][3] : 0.f;
              *reinterpret_cast<f32x4*>(stg + r * 36 + 16 * n + 4 * kk) = f32x4{skn[n].x * d0, skn[n].y * d1, skn[n].z * d2, skn[n].w * d3};
              if (nt0 + n < nnt) {
                l1n[n][0] += d0;
                l1n[n][1] += d1;
                l1n[n][2] += d2;
                l1n[n][3] += d3;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private tile: LDS runs a wave's accesses in order
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const size_t suboff = ((size_t)(sub >> 1) * Win + (sub & 1)) * (size_t)ldg + 16 * nt0 + 4 * piece;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(stg + ((lane >> 3) + 8 * h) * 36 + 4 * piece);
              if (pvx[m][h] && col_ok) *reinterpret_cast<f32x4*>(Gd + pinx[m][h] * ldg + suboff) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
          if (nt0 + n < nnt) {
            const int k4 = 16 * (nt0 + n) + 4 * kk;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float t1 = eml::row16_sum(l1n[n][g]);
              if (r == 0) my[2 * (k4 + g)] += (double)t1;
            }
          }
        }
      } else {
      // epilogue: lane owns channels k4..k4+3 of the 4 input pixels under each of its MT pooled pixels
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        const int nt = nt0 + n;
        if (nt < nnt) {  // block-uniform
          const int k4 = 16 * nt + 4 * kk;
          const float4 sk = *reinterpret_cast<const float4*>(vec_l + k4);
          const float4 tk = *reinterpret_cast<const float4*>(vec_l + Kp + k4);
          const float4 mu = *reinterpret_cast<const float4*>(vec_l + 2 * Kp + k4);
          const float4 is = *reinterpret_cast<const float4*>(vec_l + 3 * Kp + k4);
          float l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float da0 = 0.25f * acc[m][n][0], da1 = 0.25f * acc[m][n][1];
            const float da2 = 0.25f * acc[m][n][2], da3 = 0.25f * acc[m][n][3];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
              float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
              if constexpr (ACC) g = gs[m][n][sub];
              float d0, d1, d2, d3;
              float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
              if constexpr (MASKED) {
                const unsigned b = pv[m] ? mk[m][n] >> (4 * sub) : 0u;
                d0 = (b & 1u) ? da0 : 0.f;
                d1 = (b & 2u) ? da1 : 0.f;
                d2 = (b & 4u) ? da2 : 0.f;
                d3 = (b & 8u) ? da3 : 0.f;
              } else {
                x = xv[m][n][sub];
                d0 = (pv[m] && fmaf(x.x, sk.x, tk.x) > 0.f) ? da0 : 0.f;
                d1 = (pv[m] && fmaf(x.y, sk.y, tk.y) > 0.f) ? da1 : 0.f;
                d2 = (pv[m] && fmaf(x.z, sk.z, tk.z) > 0.f) ? da2 : 0.f;
                d3 = (pv[m] && fmaf(x.w, sk.w, tk.w) > 0.f) ? da3 : 0.f;
              }
              g.x = fmaf(sk.x, d0, g.x);
              g.y = fmaf(sk.y, d1, g.y);
              g.z = fmaf(sk.z, d2, g.z);
              g.w = fmaf(sk.w, d3, g.w);
              if (pv[m]) {
                const size_t pi = pin[m] + (sub >> 1) * (size_t)Win + (sub & 1);
                *reinterpret_cast<float4*>(Gd + pi * ldg + k4) = g;
              }
              l1[0] += d0;
              l1[1] += d1;
              l1[2] += d2;
              l1[3] += d3;
              if constexpr (!MASKED) {
                l2[0] = fmaf(d0, (x.x - mu.x) * is.x, l2[0]);
                l2[1] = fmaf(d1, (x.y - mu.y) * is.y, l2[1]);
                l2[2] = fmaf(d2, (x.z - mu.z) * is.z, l2[2]);
                l2[3] = fmaf(d3, (x.w - mu.w) * is.w, l2[3]);
              }
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            l1[g] = eml::row16_sum(l1[g]);
            if constexpr (!MASKED) l2[g] = eml::row16_sum(l2[g]);
            if (r == 0) {
              my[2 * (k4 + g)] += (double)l1[g];
              if constexpr (!MASKED) my[2 * (k4 + g) + 1] += (double)l2[g];
            }
          }
        }
      }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < Kp * 2; e += 256)
    partials[(size_t)blockIdx.x * Kp * 2 + e] =
        (sacc[e] + sacc[(size_t)Kp * 2 + e]) + (sacc[(size_t)2 * Kp * 2 + e] + sacc[(size_t)3 * Kp * 2 + e]);
}

// ------------------------------------------------------------------------------- dense layers: 1 or 2 layers per pass
// The dense-layer backward is HBM-bound on the O(L^2) traffic of X[:, :Cin] and G[:, :Cin].  Two
// consecutive layers (l, l-1) touch the same channels [0, Cin_{l-1}), so their dgrad is done in ONE
// pass that reads X once and read-modify-writes G once:  G += scale1_l*dam_l + scale1_{l-1}*dam_{l-1}.
// Layer l-1's dz needs the finished gradient of its 12 output channels, which needs layer l's
// contribution to exactly those channels first: that is a NARROW pass of this same kernel (NL = 1,
// channel range [Cin_{l-1}, Cin_l)).  Channels outside [k_lo, k_hi) are computed (whole 16-channel
// MFMA tiles) but contribute 0 to G and to the statistics.
struct BwdLayer {
  const float* DZ;
  const float* Zr;
  const float* cA;
  const float* cB;
  const float* cC;      // dz = cA*DZ + cB*Zr + cC  (48 channels)
  const float* Wd;      // [Kp/16][3][4][16][4]
  const float* scale1;
  const float* shift1;  // BN1 affine of this layer (Kp, zero-padded)
  double* partials;     // [grid][Kp][2]
  int Kp;
};

// MASKED: the ReLU mask of every layer comes from the bits the forward stored (BwdLayer::mask); X, mean, istd and the
// layers' shift1 are not read at all and only S1 = sum dam is accumulated (BN1's S2 follows from the weight gradient,
// bn_bwd_finalize_kernel) -- the pass moves old G, new G and dz only: (2k + 96) instead of (3k + 96) floats per pixel.
// TOP (round 6): the pass's top 24 channels [k_hi - 24, k_hi) -- the output channels of the NEXT pair of layers, whose
// backward reads them next and nobody else ever again -- leave as two compact (P, 12) tensors instead of going back into
// the wide G rows: Top[0] = channels [k_hi - 24, k_hi - 12) (the lower layer of that pair: the narrow pass's G operand),
// Top[1] = [k_hi - 12, k_hi) (the upper layer: conv3x3_bwd's g).  A 48-byte slice of a 896-byte row costs its reader one
// or two whole 128-byte lines per pixel (FETCH_SIZE per layer: 128 / 256 bytes fetched for 48 used,
// profiles/r06_perlayer_1x1.txt); a compact tensor is read line for line.  Same values, same bytes written.
template <int NL, int MT /* 16-pixel tiles per wave */, bool RAW = false /* DZ already holds dz (materialised) */,
          bool MASKED = false, bool TOP = false>
__global__ __launch_bounds__(256, 2) void conv1x1_bwd_data_multi_kernel(BwdLayer L0, BwdLayer L1,
                                                                        const float* __restrict__ X, int ldx,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ istd, int P,
                                                                        int k_lo, int k_hi, float* __restrict__ Gd,
                                                                        int ldg, int KpMax,
                                                                        const unsigned long long* __restrict__ M0,
                                                                        const unsigned long long* __restrict__ M1,
                                                                        float* __restrict__ Top) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* sacc = reinterpret_cast<double*>(smem);                         // [NL][4 waves][KpMax][2]
  float* vec_l = reinterpret_cast<float*>(sacc + (size_t)NL * 4 * KpMax * 2);  // [2 + 2*NL][KpMax]: mean, istd, (scale1, shift1) per layer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const BwdLayer Ls[2] = {L0, L1};
  for (int e = tid; e < NL * 4 * KpMax * 2; e += 256) sacc[e] = 0.0;
  for (int e = tid; e < KpMax; e += 256) {
    vec_l[e] = MASKED ? 0.f : mean[e];
    vec_l[KpMax + e] = MASKED ? 0.f : istd[e];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      vec_l[(2 + 2 * j) * KpMax + e] = e < Ls[j].Kp ? Ls[j].scale1[e] : 0.f;
      vec_l[(3 + 2 * j) * KpMax + e] = e < Ls[j].Kp ? Ls[j].shift1[e] : 0.f;
    }
  }
  __syncthreads();
  constexpr int TP = 64 * MT;  // pixels per workgroup tile
  const int ntiles = (P + TP - 1) / TP;
  const int nt_lo = k_lo >> 4, nt_hi = (k_hi + 15) >> 4;
  constexpr int NCH = 2;

  // Weight fragments of step (chunk, layer) are requested one step ahead of their MFMAs (they come from L2: issued
  // just in time, every step began with an exposed round trip).  Two register sets alternate.
  float4 wq[2][3][NCH];
  auto load_w = [&](int j, int nt0, float4 (&dst)[3][NCH]) {
    const int nnt = Ls[j].Kp >> 4;
#pragma unroll
    for (int jo = 0; jo < 3; ++jo)
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        const int nt = min(nt0 + n, nnt - 1);
        dst[jo][n] = *reinterpret_cast<const float4*>(Ls[j].Wd + ((((size_t)nt * 3 + jo) * 4 + kk) * 16 + r) * 4);
      }
  };
  load_w(0, nt_lo, wq[0]);
  int wsel = 0;  // register set holding the weights of the step about to run (compile-time after unrolling for NL = 2)

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p0 = tile * TP + wave * 16 * MT;
    long prow[MT];
    bool pv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      pv[m] = p0 + 16 * m + r < P;
      prow[m] = min(p0 + 16 * m + r, P - 1);
    }
    unsigned long long pvb[MT];   // MASKED: pixel validity as a lane mask (ANDed into the ReLU words on the scalar unit)
#pragma unroll
    for (int m = 0; m < MT; ++m) pvb[m] = MASKED ? __ballot(pv[m]) : 0ull;
    // dz fragments of every layer stay in registers for the whole tile
    float4 dz[NL][3][MT];
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
      for (int jo = 0; jo < 3; ++jo) {
        const int ch = 16 * jo + 4 * kk;
        if constexpr (RAW) {
#pragma unroll
          for (int m = 0; m < MT; ++m) dz[j][jo][m] = *reinterpret_cast<const float4*>(Ls[j].DZ + prow[m] * 48 + ch);
        } else {
          const float4 a4 = *reinterpret_cast<const float4*>(Ls[j].cA + ch);
          const float4 b4 = *reinterpret_cast<const float4*>(Ls[j].cB + ch);
          const float4 c4 = *reinterpret_cast<const float4*>(Ls[j].cC + ch);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float4 dy = *reinterpret_cast<const float4*>(Ls[j].DZ + prow[m] * 48 + ch);
            const float4 zr = *reinterpret_cast<const float4*>(Ls[j].Zr + prow[m] * 48 + ch);
            dz[j][jo][m].x = fmaf(a4.x, dy.x, fmaf(b4.x, zr.x, c4.x));
            dz[j][jo][m].y = fmaf(a4.y, dy.y, fmaf(b4.y, zr.y, c4.y));
            dz[j][jo][m].z = fmaf(a4.z, dy.z, fmaf(b4.z, zr.z, c4.z));
            dz[j][jo][m].w = fmaf(a4.w, dy.w, fmaf(b4.w, zr.w, c4.w));
          }
        }
      }
    for (int nt0 = nt_lo; nt0 < nt_hi; nt0 += NCH) {
      // X and the old G of this chunk are requested now and consumed after the MFMAs (latency hidden)
      // (lanes whose 4 channels lie entirely outside [k_lo, k_hi) re-read the range's first quad -- an L1 hit --
      // instead of pulling unused channels from HBM, and do not write back)
      float4 gs[MT][NCH], xv[MT][NCH];
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        const int k4 = 16 * min(nt0 + n, nt_hi - 1) + 4 * kk;
        const int k4l = (k4 + 3 >= k_lo && k4 < k_hi) ? k4 : (k_lo & ~3);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          gs[m][n] = *reinterpret_cast<const float4*>(Gd + prow[m] * ldg + k4l);
          if constexpr (!MASKED) xv[m][n] = *reinterpret_cast<const float4*>(X + prow[m] * ldx + k4l);
        }
      }
      const int nt_next = nt0 + NCH < nt_hi ? nt0 + NCH : nt_lo;  // the next tile starts at nt_lo again
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int nnt = Ls[j].Kp >> 4;
        // MASKED: the forward's ballot words of this step's (pixel group, channel tile)s -- wave-uniform addresses of a
        // read-only buffer, i.e. scalar loads; requested here, used after the step's MFMAs
        unsigned long long mk[MT][NCH][4];
        if constexpr (MASKED) {
          const unsigned long long* __restrict__ Mj = j == 0 ? M0 : M1;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NCH; ++n) {
              const size_t wi =
                  ((size_t)__builtin_amdgcn_readfirstlane((p0 >> 4) + m) * nnt + min(nt0 + n, nnt - 1)) * 4;
#pragma unroll
              for (int g = 0; g < 4; ++g) mk[m][n][g] = Mj[wi + g];
            }
        }
        // request the NEXT step's weights, then run this step
        const int cur = NL == 2 ? j : wsel;
        if (j + 1 < NL)
          load_w(j + 1, nt0, wq[cur ^ 1]);
        else
          load_w(0, nt_next, wq[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[MT][NCH];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NCH; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jo = 0; jo < 3; ++jo)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < NCH; ++n)
#pragma unroll
              for (int m = 0; m < MT; ++m) acc[m][n] = mfma16(f4c(wq[cur][jo][n], t), f4c(dz[j][jo][m], t), acc[m][n]);
        if (NL == 1) wsel ^= 1;
        double* my = sacc + ((size_t)(j * 4 + wave) * KpMax) * 2;
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
          const int nt = nt0 + n;
          if (nt < nnt && nt < nt_hi) {  // block-uniform
            const int k4 = 16 * nt + 4 * kk;
            const float4 mu = *reinterpret_cast<const float4*>(vec_l + k4);
            const float4 is = *reinterpret_cast<const float4*>(vec_l + KpMax + k4);
            const float4 sk = *reinterpret_cast<const float4*>(vec_l + (2 + 2 * j) * KpMax + k4);
            const float4 tk = *reinterpret_cast<const float4*>(vec_l + (3 + 2 * j) * KpMax + k4);
            const bool in0 = k4 + 0 >= k_lo && k4 + 0 < k_hi, in1 = k4 + 1 >= k_lo && k4 + 1 < k_hi;
            const bool in2 = k4 + 2 >= k_lo && k4 + 2 < k_hi, in3 = k4 + 3 >= k_lo && k4 + 3 < k_hi;
            unsigned long long inb[4] = {0ull, 0ull, 0ull, 0ull};
            if constexpr (MASKED) {
              inb[0] = __ballot(in0);
              inb[1] = __ballot(in1);
              inb[2] = __ballot(in2);
              inb[3] = __ballot(in3);
            }
            float l1[4] = {0.f, 0.f, 0.f, 0.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              float d0, d1, d2, d3;
              if constexpr (MASKED) {
                // bit `lane` of word g = this lane's ReLU mask for channel g: the word IS the select mask of v_cndmask
                d0 = __builtin_amdgcn_inverse_ballot_w64(mk[m][n][0] & pvb[m] & inb[0]) ? acc[m][n][0] : 0.f;
                d1 = __builtin_amdgcn_inverse_ballot_w64(mk[m][n][1] & pvb[m] & inb[1]) ? acc[m][n][1] : 0.f;
                d2 = __builtin_amdgcn_inverse_ballot_w64(mk[m][n][2] & pvb[m] & inb[2]) ? acc[m][n][2] : 0.f;
                d3 = __builtin_amdgcn_inverse_ballot_w64(mk[m][n][3] & pvb[m] & inb[3]) ? acc[m][n][3] : 0.f;
              } else {
                // unconditional use of the loaded x / g (masked values): see conv3x3_bwd_data_kernel
                const float4 x = xv[m][n];
                d0 = (pv[m] && in0 && fmaf(x.x, sk.x, tk.x) > 0.f) ? acc[m][n][0] : 0.f;
                d1 = (pv[m] && in1 && fmaf(x.y, sk.y, tk.y) > 0.f) ? acc[m][n][1] : 0.f;
                d2 = (pv[m] && in2 && fmaf(x.z, sk.z, tk.z) > 0.f) ? acc[m][n][2] : 0.f;
                d3 = (pv[m] && in3 && fmaf(x.w, sk.w, tk.w) > 0.f) ? acc[m][n][3] : 0.f;
                l2[0] = fmaf(d0, (x.x - mu.x) * is.x, l2[0]);
                l2[1] = fmaf(d1, (x.y - mu.y) * is.y, l2[1]);
                l2[2] = fmaf(d2, (x.z - mu.z) * is.z, l2[2]);
                l2[3] = fmaf(d3, (x.w - mu.w) * is.w, l2[3]);
              }
              gs[m][n].x = fmaf(sk.x, d0, gs[m][n].x);
              gs[m][n].y = fmaf(sk.y, d1, gs[m][n].y);
              gs[m][n].z = fmaf(sk.z, d2, gs[m][n].z);
              gs[m][n].w = fmaf(sk.w, d3, gs[m][n].w);
              l1[0] += d0;
              l1[1] += d1;
              l1[2] += d2;
              l1[3] += d3;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              l1[g] = eml::row16_sum(l1[g]);  // over the 16 pixels (lanes r) of this lane row: DPP, no LDS
              if constexpr (!MASKED) l2[g] = eml::row16_sum(l2[g]);
              if (r == 0) {
                my[2 * (k4 + g)] += (double)l1[g];
                if constexpr (!MASKED) my[2 * (k4 + g) + 1] += (double)l2[g];
              }
            }
          }
        }
      }
      // one read-modify-write of G for all layers of the pass
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        const int nt = nt0 + n;
        if (nt < nt_hi) {
          const int k4 = 16 * nt + 4 * kk;
          const bool act = k4 + 3 >= k_lo && k4 < k_hi;
          if constexpr (TOP) {
            // quads of the top 24 channels go to the compact tensors (k_hi is a multiple of 4: a quad lies on one side)
            const int t = k4 - (k_hi - 24);
            float* cdst = Top + (t >= 12 ? (size_t)P * 12 + (t - 12) : (size_t)max(t, 0));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              float* dst = t >= 0 ? cdst + prow[m] * 12 : Gd + prow[m] * ldg + k4;
              if (pv[m] && act) *reinterpret_cast<float4*>(dst) = gs[m][n];
            }
          } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              if (pv[m] && act) *reinterpret_cast<float4*>(Gd + prow[m] * ldg + k4) = gs[m][n];  // gs = old G + updates
            }
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int Kp = Ls[j].Kp;
    const double* base = sacc + (size_t)j * 4 * KpMax * 2;
    for (int e = tid; e < Kp * 2; e += 256)
      Ls[j].partials[(size_t)blockIdx.x * Kp * 2 + e] = (base[e] + base[(size_t)KpMax * 2 + e]) +
                                                        (base[(size_t)2 * KpMax * 2 + e] + base[(size_t)3 * KpMax * 2 + e]);
  }
}

// =============================================================================== conv1x1 backward: narrow data pass
// The data gradient of layer l over the 12 output channels of layer l-1 -- [k_lo, k_lo+12) of the block buffer -- has
// to be finished before layer l-1's own backward can start.  It is a 48 -> 12 product per pixel, so it gets its own
// streaming kernel on compact operands: dz (P,48, materialised by the weight-gradient kernel) in, the increment
// N12[p][c] = G[p][k_lo+c] + scale1[k_lo+c] * mask * sum_o dz[p][o] W1[o][k_lo+c], i.e. the FINISHED gradient of those
// 12 channels, out as a compact (P,12) tensor that the lower layer's conv3x3_bwd_data stages instead of the G slice
// (the wide G rows are read once here for 48 bytes and never written).
// D^T form: A = W1 (rows = 12 channels of 16, k = o), B = dz (k = o, columns = 16 pixels); the k index is a
// summation label, so lane (r, kk) feeds dz[p][16jo + 4kk + t] at step (jo, t) straight from its float4 load.
// Statistics (sum dam, sum dam*xhat) accumulate per element in f64.
template <int MT>
__global__ __launch_bounds__(256, 2) void conv1x1_bwd_narrow_kernel(
    const float* __restrict__ DZ, const float* __restrict__ W1 /*[48][Cin]*/, int Cin, int k_lo,
    const float* __restrict__ X, int ldx, const float* __restrict__ scale1, const float* __restrict__ shift1,
    const float* __restrict__ mean, const float* __restrict__ istd, int P, const float* __restrict__ Gd, int ldg,
    float* __restrict__ N12, double* __restrict__ partials /*[grid][Kp][2]*/, int Kp) {
  __shared__ double red[4][12][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  float aw[3][4];
#pragma unroll
  for (int jo = 0; jo < 3; ++jo)
#pragma unroll
    for (int t = 0; t < 4; ++t) aw[jo][t] = r < 12 ? W1[(size_t)(16 * jo + 4 * kk + t) * Cin + k_lo + r] : 0.f;
  const bool cv = kk < 3;                 // lanes of row group 3 hold the padding channels 12..15
  const int cq = k_lo + 4 * (cv ? kk : 2);  // first of this lane's 4 channels (clamped for the padding lanes)
  float sk[4], tk[4], mu[4], is[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sk[g] = scale1[cq + g];
    tk[g] = shift1[cq + g];
    mu[g] = mean[cq + g];
    is[g] = istd[cq + g];
  }
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int TP = 64 * MT;
  const int ntiles = (P + TP - 1) / TP;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p0 = tile * TP + wave * 16 * MT;
    size_t prow[MT];
    bool pv[MT];
    float4 dz[MT][3];
    float2 xa[MT], xb[MT], ga[MT], gb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      pv[m] = p0 + 16 * m + r < P;
      prow[m] = (size_t)min(p0 + 16 * m + r, P - 1);
#pragma unroll
      for (int jo = 0; jo < 3; ++jo) dz[m][jo] = *reinterpret_cast<const float4*>(DZ + prow[m] * 48 + 16 * jo + 4 * kk);
      xa[m] = *reinterpret_cast<const float2*>(X + prow[m] * ldx + cq);       // k_lo is even: 8-byte aligned
      xb[m] = *reinterpret_cast<const float2*>(X + prow[m] * ldx + cq + 2);
      ga[m] = *reinterpret_cast<const float2*>(Gd + prow[m] * ldg + cq);
      gb[m] = *reinterpret_cast<const float2*>(Gd + prow[m] * ldg + cq + 2);
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jo = 0; jo < 3; ++jo)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma16(aw[jo][t], f4c(dz[m][jo], t), acc[m]);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float x[4] = {xa[m].x, xa[m].y, xb[m].x, xb[m].y};
      const float gold[4] = {ga[m].x, ga[m].y, gb[m].x, gb[m].y};
      float o[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float d = (pv[m] && cv && fmaf(x[g], sk[g], tk[g]) > 0.f) ? acc[m][g] : 0.f;
        o[g] = fmaf(sk[g], d, gold[g]);
        s1[g] += (double)d;
        s2[g] += (double)(d * ((x[g] - mu[g]) * is[g]));
      }
      if (pv[m] && cv) *reinterpret_cast<float4*>(N12 + prow[m] * 12 + 4 * kk) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      s1[g] += shfl_xor_d(s1[g], o);
      s2[g] += shfl_xor_d(s2[g], o);
    }
    if (r == 0 && cv) {
      red[wave][4 * kk + g][0] = s1[g];
      red[wave][4 * kk + g][1] = s2[g];
    }
  }
  __syncthreads();
  if (tid < 24) {
    const int c = tid >> 1, e = tid & 1;
    partials[((size_t)blockIdx.x * Kp + k_lo + c) * 2 + e] = (red[0][c][e] + red[1][c][e]) + (red[2][c][e] + red[3][c][e]);
  }
}

// Wd[nt][jo][kk][col][t] = W[o = 16jo+4kk+t][k = 16nt+col]  (W is [Cout][Cin]); zero outside.
__global__ void permute_w1_bwd_kernel(const float* __restrict__ W, int Cout, int Cin, int Kp, int Ko,
                                      float* __restrict__ Wd) {
  const size_t total = (size_t)Kp * Ko;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(e & 3), col = (int)((e >> 2) & 15), kk = (int)((e >> 6) & 3);
    const size_t rest = e >> 8;
    const int jo = (int)(rest % (Ko >> 4)), nt = (int)(rest / (Ko >> 4));
    const int o = 16 * jo + 4 * kk + t, k = 16 * nt + col;
    Wd[e] = (o < Cout && k < Cin) ? W[(size_t)o * Cin + k] : 0.f;
  }
}

// =============================================================================== deferred BN1 affine
// Gfull[p][c] = G[p][c] + sB[c]*X[p][c] + sC[c] for the n channels [c0, c0+n) whose gradient is now
// complete (every later layer has added its scale1*dam term and its (cB, cC) into sB/sC).
__global__ __launch_bounds__(256) void grad_materialize_kernel(float* __restrict__ Gd, int ldg,
                                                               const float* __restrict__ X, int ldx,
                                                               const float* __restrict__ sB,
                                                               const float* __restrict__ sC, int c0, int n, size_t P) {
  const int nh = n >> 1;
  const size_t total = P * nh;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t p = e / nh;
    const int c = c0 + 2 * (int)(e - p * nh);
    const float2 x = *reinterpret_cast<const float2*>(X + p * ldx + c);
    float2* gp = reinterpret_cast<float2*>(Gd + p * ldg + c);
    float2 g = *gp;
    g.x += fmaf(sB[c], x.x, sC[c]);
    g.y += fmaf(sB[c + 1], x.y, sC[c + 1]);
    *gp = g;
  }
}

// =============================================================================== BN backward: stats (elementwise BN)
// For y = act(BN(raw)):  dym = dy * (relu ? out > 0 : 1);  S1 = sum dym, S2 = sum dym * rawhat.
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ DY, int ld_dy,
                                                           const float* __restrict__ raw, int ld_raw,
                                                           const float* __restrict__ out, int ld_out, int relu, int C,
                                                           size_t P, const float* __restrict__ mean,
                                                           const float* __restrict__ istd,
                                                           double* __restrict__ partials /*[grid][C][2]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* red = reinterpret_cast<double*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = (C + 63) >> 6;
  float mu[6], is[6];
  double a1[6], a2[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int c = lane + 64 * k;
    mu[k] = (k < nk && c < C) ? mean[c] : 0.f;
    is[k] = (k < nk && c < C) ? istd[c] : 0.f;
    a1[k] = a2[k] = 0.0;
  }
  // kUN pixels per wave and iteration, all loads issued before the first use (see bn_apply_kernel)
  constexpr int kUN = 4;
  for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * kUN; p0 < P; p0 += (size_t)gridDim.x * 4 * kUN) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k < nk) {  // wave-uniform
        const int c = min(lane + 64 * k, C - 1);
        float dy[kUN], rw[kUN], ou[kUN];
#pragma unroll
        for (int u = 0; u < kUN; ++u) {
          const size_t pu = (p0 + u < P) ? p0 + u : P - 1;
          dy[u] = DY[pu * ld_dy + c];
          rw[u] = raw[pu * ld_raw + c];
          ou[u] = relu ? out[pu * ld_out + c] : 1.f;
        }
        // per-element f64 accumulation (the loads above are batched; only the converts / adds are per element)
        const bool cv = lane + 64 * k < C;
#pragma unroll
        for (int u = 0; u < kUN; ++u) {
          const float d = (cv && p0 + u < P && ou[u] > 0.f) ? dy[u] : 0.f;
          a1[k] += (double)d;
          a2[k] += (double)(d * ((rw[u] - mu[k]) * is[k]));
        }
      }
    }
  }
  const int CP = nk * 64;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (k < nk) {
      red[(wave * CP + lane + 64 * k) * 2 + 0] = a1[k];
      red[(wave * CP + lane + 64 * k) * 2 + 1] = a2[k];
    }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256)
    partials[(size_t)blockIdx.x * C * 2 + e] =
        (red[e] + red[CP * 2 + e]) + (red[2 * CP * 2 + e] + red[3 * CP * 2 + e]);
}

// The same sums for FEW channels (C <= 32, a multiple of 4: norm0's 24 at the full resolution; see bn_apply_small_kernel): a
// lane owns 4 consecutive channels of a pixel, 16-byte loads, 64 / (C / 4) pixels per wave instruction.
__global__ __launch_bounds__(256) void bn_bwd_stats_small_kernel(const float* __restrict__ DY, int ld_dy,
                                                                 const float* __restrict__ raw, int ld_raw,
                                                                 const float* __restrict__ out, int ld_out, int relu, int C,
                                                                 size_t P, const float* __restrict__ mean,
                                                                 const float* __restrict__ istd,
                                                                 double* __restrict__ partials /*[grid][C][2]*/) {
  __shared__ double red[4][64][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CQ = C >> 2, PW = 64 / CQ;
  const int slot = lane / CQ, quad = lane - slot * CQ;
  const bool live = slot < PW;
  const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * quad), is = *reinterpret_cast<const float4*>(istd + 4 * quad);
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int kUN = 4;
  const size_t step = (size_t)gridDim.x * 4 * kUN * PW;
  for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * kUN * PW; p0 < P; p0 += step) {
    float4 dy[kUN], rw[kUN], ou[kUN];
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const size_t p = p0 + (size_t)u * PW + slot;
      const size_t pc = p < P ? p : P - 1;
      dy[u] = *reinterpret_cast<const float4*>(DY + pc * ld_dy + 4 * quad);
      rw[u] = *reinterpret_cast<const float4*>(raw + pc * ld_raw + 4 * quad);
      ou[u] = relu ? *reinterpret_cast<const float4*>(out + pc * ld_out + 4 * quad) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const bool ok = live && p0 + (size_t)u * PW + slot < P;
      const float d[4] = {(ok && ou[u].x > 0.f) ? dy[u].x : 0.f, (ok && ou[u].y > 0.f) ? dy[u].y : 0.f,
                          (ok && ou[u].z > 0.f) ? dy[u].z : 0.f, (ok && ou[u].w > 0.f) ? dy[u].w : 0.f};
      const float xh[4] = {(rw[u].x - mu.x) * is.x, (rw[u].y - mu.y) * is.y, (rw[u].z - mu.z) * is.z, (rw[u].w - mu.w) * is.w};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        a1[g] += (double)d[g];
        a2[g] += (double)(d[g] * xh[g]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    red[wave][lane][2 * g] = live ? a1[g] : 0.0;
    red[wave][lane][2 * g + 1] = live ? a2[g] : 0.0;
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

// norm0 + relu0 backward statistics with the block's deferred BN1 affine folded in and WITHOUT the block buffer (round 6): the
// incoming gradient of block 1's C0 input channels is G + sB*x + sC with x = relu(scale0*y0 + shift0) -- the value the forward's
// bn_apply wrote into the block buffer, recomputed from the raw conv0 output y0 with the same expression -- so neither the
// grad_materialize pass over G (a read-modify-write of 96 bytes per 896-byte row) nor the read of X's first line happens.
__global__ __launch_bounds__(256) void norm0_bwd_stats_kernel(const float* __restrict__ Gd, int ldg, const float* __restrict__ Y0,
                                                              int ld_y0, const float* __restrict__ scale0,
                                                              const float* __restrict__ shift0, const float* __restrict__ sB,
                                                              const float* __restrict__ sC, int C, size_t P,
                                                              const float* __restrict__ mean, const float* __restrict__ istd,
                                                              double* __restrict__ partials /*[grid][C][2]*/) {
  __shared__ double red[4][64][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CQ = C >> 2, PW = 64 / CQ;
  const int slot = lane / CQ, quad = lane - slot * CQ;
  const bool live = slot < PW;
  auto ld4 = [&](const float* q) { return *reinterpret_cast<const float4*>(q + 4 * quad); };
  const float4 mu = ld4(mean), is = ld4(istd), s0 = ld4(scale0), t0 = ld4(shift0), b4 = ld4(sB), c4 = ld4(sC);
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int kUN = 4;
  const size_t step = (size_t)gridDim.x * 4 * kUN * PW;
  for (size_t p0 = ((size_t)blockIdx.x * 4 + wave) * kUN * PW; p0 < P; p0 += step) {
    float4 gv[kUN], yv[kUN];
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const size_t p = p0 + (size_t)u * PW + slot;
      const size_t pc = p < P ? p : P - 1;
      gv[u] = *reinterpret_cast<const float4*>(Gd + pc * ldg + 4 * quad);
      yv[u] = *reinterpret_cast<const float4*>(Y0 + pc * ld_y0 + 4 * quad);
    }
#pragma unroll
    for (int u = 0; u < kUN; ++u) {
      const bool ok = live && p0 + (size_t)u * PW + slot < P;
      const float y[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w}, g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      const float sc[4] = {s0.x, s0.y, s0.z, s0.w}, sh[4] = {t0.x, t0.y, t0.z, t0.w};
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, cc[4] = {c4.x, c4.y, c4.z, c4.w};
      const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ii[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = fmaxf(fmaf(y[e], sc[e], sh[e]), 0.f);            // bn_apply's expression
        const float dy = g[e] + fmaf(bb[e], x, cc[e]);                   // grad_materialize's expression
        const float d = (ok && x > 0.f) ? dy : 0.f;
        a1[e] += (double)d;
        a2[e] += (double)(d * ((y[e] - mm[e]) * ii[e]));
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    red[wave][lane][2 * g] = live ? a1[g] : 0.0;
    red[wave][lane][2 * g + 1] = live ? a2[g] : 0.0;
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

// =============================================================================== conv0 backward: weight
// dW0[o][c][ky][kx] = sum_p dY0[p][o] * x[b][c][y+ky-1][x+kx-1],
// dY0 = cA*(G*(X1>0)) + cB*Y0 + cC (norm0 + relu0 backward folded into the operand).
// D[i=(c,ky,kx) 27->32][j=o 24->32], MFMA-k = pixel.
// FUSED (round 6): X1 is not read -- the block buffer's value is recomputed from Y0 (relu(scale0*y0 + shift0), bn_apply's
// expression) and the block's deferred BN1 affine is applied to G here (g = G + sB*x + sC, grad_materialize's expression).
template <bool FUSED>
__global__ __launch_bounds__(256) void conv0_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ Gd, int ldg, const float* __restrict__ X1, int ldx,
    const float* __restrict__ Y0, int C0, const float* __restrict__ cA, const float* __restrict__ cB,
    const float* __restrict__ cC, int B, int H, int W, float* __restrict__ partial /*[grid*4][32][32]*/,
    const float* __restrict__ scale0, const float* __restrict__ shift0, const float* __restrict__ sB,
    const float* __restrict__ sC) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ca[2], cb[2], cc[2], fs0[2] = {0.f, 0.f}, ft0[2] = {0.f, 0.f}, fb[2] = {0.f, 0.f}, fc[2] = {0.f, 0.f};
  bool ov[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    ov[n] = 16 * n + r < C0;
    ca[n] = ov[n] ? cA[16 * n + r] : 0.f;
    cb[n] = ov[n] ? cB[16 * n + r] : 0.f;
    cc[n] = ov[n] ? cC[16 * n + r] : 0.f;
    if constexpr (FUSED) {
      fs0[n] = ov[n] ? scale0[16 * n + r] : 0.f;
      ft0[n] = ov[n] ? shift0[16 * n + r] : 0.f;
      fb[n] = ov[n] ? sB[16 * n + r] : 0.f;
      fc[n] = ov[n] ? sC[16 * n + r] : 0.f;
    }
  }
  int ci[2], dyi[2], dxi[2];
  bool iv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = 16 * i + r;
    iv[i] = t < 27;
    ci[i] = iv[i] ? t / 9 : 0;
    dyi[i] = iv[i] ? (t % 9) / 3 - 1 : 0;
    dxi[i] = iv[i] ? (t % 3) - 1 : 0;
  }
  const size_t P = (size_t)B * H * W, plane = (size_t)H * W;
  const size_t nq = (P + 3) >> 2;
  // kU pixel quads per turn, every load of the turn issued (unconditionally, from clamped addresses) before the first is used
  // (round 6): one quad at a time was a dependent load -> 4 MFMAs chain per iteration, 600 iterations per wave: 0.63 ms of latency
  constexpr int kU = 4;
  for (size_t q0 = ((size_t)blockIdx.x * 4 + wave) * kU; q0 < nq; q0 += (size_t)gridDim.x * 4 * kU) {
    float av[kU][2], gv[kU][2], xv[kU][2], yv[kU][2];
    bool pvu[kU], inb[kU][2];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const size_t p = 4 * (q0 + u) + kk;
      pvu[u] = p < P;
      const size_t pc = pvu[u] ? p : P - 1;
      const int b = (int)(pc / plane), rem = (int)(pc - (size_t)b * plane);
      const int yy = rem / W, xx = rem - yy * W;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int gy = yy + dyi[i], gx = xx + dxi[i];
        inb[u][i] = pvu[u] && iv[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gx, 0), W - 1);
        av[u][i] = x[((size_t)(b * 3 + ci[i]) * H + gyc) * W + gxc];
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int o = ov[n] ? 16 * n + r : 0;
        gv[u][n] = Gd[pc * ldg + o];
        if constexpr (!FUSED) xv[u][n] = X1[pc * ldx + o];
        yv[u][n] = Y0[pc * C0 + o];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      float a[2], d[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = inb[u][i] ? av[u][i] : 0.f;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float g;
        if constexpr (FUSED) {
          const float xx = fmaxf(fmaf(yv[u][n], fs0[n], ft0[n]), 0.f);
          g = xx > 0.f ? gv[u][n] + fmaf(fb[n], xx, fc[n]) : 0.f;
        } else {
          g = xv[u][n] > 0.f ? gv[u][n] : 0.f;
        }
        const float v = fmaf(ca[n], g, fmaf(cb[n], yv[u][n], cc[n]));
        d[n] = (pvu[u] && ov[n]) ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = mfma16(a[i], d[n], acc[i][n]);
    }
  }
  float* out = partial + ((size_t)blockIdx.x * 4 + wave) * 1024;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) out[(16 * i + 4 * kk + g) * 32 + 16 * n + r] = acc[i][n][g];
}

// =============================================================================== head backward
// dF[p][c] = (F[p][c] > 0) ? gpooled[b][c][oy/k][ox/k] / k^2 : 0
__global__ __launch_bounds__(256) void head_pool_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ F,
                                                            int ldf, int C, int B, int H, int W, int k,
                                                            float* __restrict__ dF, int ldd) {
  const int Ho = H / k, Wo = W / k;
  const size_t total = (size_t)B * H * W * C;
  const float inv = 1.0f / (float)(k * k);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const size_t p = e / C;
    const int xx = (int)(p % W);
    const size_t rest = p / W;
    const int yy = (int)(rest % H), b = (int)(rest / H);
    const int oy = yy / k, ox = xx / k;
    float v = 0.f;
    if (oy < Ho && ox < Wo && F[p * ldf + c] > 0.f) v = gp[(((size_t)b * C + c) * Ho + oy) * Wo + ox] * inv;
    dF[p * ldd + c] = v;
  }
}

}  // namespace

// =============================================================================== BN1 dgamma, direct (ill-conditioned channels)
// For the channels bn_bwd_finalize_kernel flagged: S2[c] = sum_p dy[p][c] * xhat[p][c] accumulated per element in f64, with
// dy = relu-mask * (sum_o W[o][c] dz[p][o]) (pool: the pooled pixel's dz spread over its 2x2 window, / 4) -- the definition
// the x-free passes replaced by the weight-gradient identity.  A rare path (no flag: both kernels return at once): one
// flagged channel at a time, every thread striding over the block's pixel range; partial [grid][Cin] doubles.
__global__ __launch_bounds__(256) void bn_dgamma_direct_partial_kernel(
    const float* __restrict__ X, int ldx, long P, int Hin, int Win, int pool, const float* __restrict__ DY, int ld_dy,
    const float* __restrict__ Zr, int ld_z, const float* __restrict__ cA, const float* __restrict__ cB,
    const float* __restrict__ cC, int Cout, const float* __restrict__ W, int Cin, const float* __restrict__ scale1,
    const float* __restrict__ shift1, const float* __restrict__ mean, const float* __restrict__ istd,
    const int* __restrict__ cond, double* __restrict__ partial) {
  __shared__ int list[384];
  __shared__ int cnt;
  __shared__ float wcol[192], co[3][192];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (int c = tid; c < Cin; c += 256)
    if (cond[c]) list[atomicAdd(&cnt, 1)] = c;   // order irrelevant: every channel's sum is its own
  __syncthreads();
  const int n = cnt;
  if (n == 0) return;
  for (int o = tid; o < Cout; o += 256) {
    co[0][o] = Zr ? cA[o] : 1.f;
    co[1][o] = Zr ? cB[o] : 0.f;
    co[2][o] = Zr ? cC[o] : 0.f;
  }
  const long per = (P + gridDim.x - 1) / gridDim.x, p_lo = (long)blockIdx.x * per, p_hi = min(P, p_lo + per);
  const int Ho = Hin >> 1, Wo = Win >> 1;
  for (int li = 0; li < n; ++li) {
    const int c = list[li];
    __syncthreads();
    for (int o = tid; o < Cout; o += 256) wcol[o] = W[(size_t)o * Cin + c];
    __syncthreads();
    const float s1 = scale1[c], t1 = shift1[c];
    const double mu = mean[c], is = istd[c];
    double acc = 0.0;
    for (long p = p_lo + tid; p < p_hi; p += 256) {
      const float x = X[(size_t)p * ldx + c];
      if (fmaf(x, s1, t1) > 0.f) {
        long pp = p;
        if (pool) {
          const long b = p / ((long)Hin * Win), rem = p - b * (long)Hin * Win;
          const int y = (int)(rem / Win), xx = (int)(rem - (long)y * Win);
          pp = (b * Ho + (y >> 1)) * Wo + (xx >> 1);
        }
        const float* dy = DY + (size_t)pp * ld_dy;
        const float* zr = Zr ? Zr + (size_t)pp * ld_z : dy;
        double d = 0.0;
        for (int o = 0; o < Cout; ++o) d += (double)wcol[o] * (double)fmaf(co[0][o], dy[o], fmaf(co[1][o], zr[o], co[2][o]));
        if (pool) d *= 0.25;
        acc += d * (((double)x - mu) * is);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_d(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[(size_t)blockIdx.x * Cin + c] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ __launch_bounds__(256) void bn_dgamma_direct_reduce_kernel(const double* __restrict__ partial, int R, int Cin,
                                                                      const int* __restrict__ cond,
                                                                      float* __restrict__ dgamma) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Cin || !cond[c]) return;
  double s = 0.0;
  for (int g = 0; g < R; ++g) s += partial[(size_t)g * Cin + c];
  dgamma[c] = (float)s;
}

// =============================================================================== C ABI
extern "C" int eml_dense_conv3x3_bwd_data_f32(const float* G, int ldg, int c0, const float* W2, const float* Z,
                                              const float* zmean, const float* zistd, float* DZ, int B, int H, int W,
                                              double* partials, int grid, const float* X, int ldx, int cx,
                                              const float* sB, const float* sC, float* GF, eml_stream_t stream) {
  if (!G || !W2 || !Z || !zmean || !zistd || !DZ || !partials || B < 1 || H < 1 || W < 1 || grid < 1 || (c0 & 1) ||
      (ldg & 1))
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_bwd_data_f32: bad arguments");
  if (X) {
    if (!sB || !sC || !GF || (ldx & 1))
      return eml::fail(EML_EINVAL, "eml_dense_conv3x3_bwd_data_f32: fused affine needs X, sB, sC, GF");
    if (cx < 0 || (cx & 1)) return eml::fail(EML_EINVAL, "eml_dense_conv3x3_bwd_data_f32: cx must be even");
  }
  // 16-byte staging where every slice it touches is 16-byte aligned (EML_D3_NARROW=1: the float2 path, for the A/B);
  // 8-row tiles / 512 threads / one workgroup per CU; EML_D3_SHORT=1: the 4-row / 256-thread / two-per-CU geometry of the
  // round-4 A/B (15.34 against 14.43 ms per step: slower, profiles/r04_ab_conv3x3.txt)
  static const bool narrow = [] { const char* v = getenv("EML_D3_NARROW"); return v && v[0] == '1'; }();
  static const bool tall = [] { const char* v = getenv("EML_D3_SHORT"); return !(v && v[0] == '1'); }();
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool wide = !narrow && (ldg & 3) == 0 && (c0 & 3) == 0 && al16(G) &&
                    (!X || ((ldx & 3) == 0 && (cx & 3) == 0 && al16(X) && al16(sB) && al16(sC) && al16(GF)));
#define EML_LAUNCH_D3(FUSEV, WIDEV, THV)                                                                                \
  hipLaunchKernelGGL((conv3x3_bwd_data_kernel<FUSEV, WIDEV, THV>), dim3(grid), dim3(THV * 64), 0, (hipStream_t)stream, G, ldg, \
                     c0, W2, Z, zmean, zistd, DZ, B, H, W, partials, X, ldx, cx, sB, sC, GF)
  if (X) {
    if (tall) { if (wide) EML_LAUNCH_D3(true, true, 8); else EML_LAUNCH_D3(true, false, 8); }
    else      { EML_LAUNCH_D3(true, false, 4); }   // (the 16-byte staging spills 18 registers next to the fused affine here)
  } else {
    if (tall) { if (wide) EML_LAUNCH_D3(false, true, 8); else EML_LAUNCH_D3(false, false, 8); }
    else      { if (wide) EML_LAUNCH_D3(false, true, 4); else EML_LAUNCH_D3(false, false, 4); }
  }
#undef EML_LAUNCH_D3
  return eml::check_launch("eml_dense_conv3x3_bwd_data_f32");
}

extern "C" int eml_dense_conv3x3_bwd_weight_f32(const float* G, int ldg, int c0, const float* Z, const float* scale2,
                                                const float* shift2, int B, int H, int W, float* partial, float* dW2,
                                                int grid, eml_stream_t stream) {
  if (!G || !Z || !scale2 || !shift2 || !partial || !dW2 || B < 1 || H < 1 || W < 1 || grid < 1)
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_bwd_weight_f32: bad arguments");
  // 16-byte loads of the 12-channel slice need ldg and c0 multiples of 4 (always true for the engine's buffers)
  static const bool greg = [] { const char* v = getenv("EML_W3_GREG"); return v && v[0] == '1'; }();
  if (!greg && (ldg & 3) == 0 && (c0 & 3) == 0 && (reinterpret_cast<uintptr_t>(G) & 15) == 0) {
    const size_t lds = (size_t)(2 * kHH * kHW * kPSW + 2 * kGL) * sizeof(float);
    EML_ENSURE_LDS((&conv3x3_bwd_weight_kernel<true>), lds);
    hipLaunchKernelGGL(conv3x3_bwd_weight_kernel<true>, dim3(grid), dim3(kBW), lds, (hipStream_t)stream, G, ldg, c0, Z,
                       scale2, shift2, B, H, W, partial);
  } else {
    const size_t lds = (size_t)(2 * kHH * kHW * kPSW) * sizeof(float);
    EML_ENSURE_LDS((&conv3x3_bwd_weight_kernel<false>), lds);
    hipLaunchKernelGGL(conv3x3_bwd_weight_kernel<false>, dim3(grid), dim3(kBW), lds, (hipStream_t)stream, G, ldg, c0, Z,
                       scale2, shift2, B, H, W, partial);
  }
  int rc = eml::check_launch("eml_dense_conv3x3_bwd_weight_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(27 * 256 / 64), dim3(256), 0, (hipStream_t)stream, partial, 2 * grid,
                     (size_t)27 * 256, 1, 0, 0, 0, dW2);
  return eml::check_launch("eml_dense_conv3x3_bwd_weight_f32(reduce)");
}

#ifdef EML_STAMPS
extern "C" int eml_c3_read_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(eml_c3_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    const unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(eml_c3_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
// One launch for the pair above (see conv3x3_bwd_fused_kernel).  Returns EML_EINVAL when the buffers do not allow the
// 16-byte staging (the caller then issues the two separate launches).
extern "C" int eml_dense_conv3x3_bwd_fused_supported(int ldg, int c0, int ldx, int cx) {
  return ((ldg & 1) == 0 && (c0 & 1) == 0 && (ldx & 1) == 0 && (cx & 1) == 0) ? 1 : 0;
}
extern "C" int eml_dense_conv3x3_bwd_fused_f32(const float* G, int ldg, int c0, const float* W2, const float* Z,
                                               const float* zmean, const float* zistd, float* DZ, int B, int H, int W,
                                               double* partials, int grid, const float* X, int ldx, int cx,
                                               const float* sB, const float* sC, float* GF, const float* scale2,
                                               const float* shift2, float* partialW, float* dW2, eml_stream_t stream) {
  if (!G || !W2 || !Z || !zmean || !zistd || !DZ || !partials || !X || !sB || !sC || !GF || !scale2 || !shift2 || !partialW ||
      !dW2 || B < 1 || H < 1 || W < 1 || grid < 1 || cx < 0)
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_bwd_fused_f32: bad arguments");
  const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!eml_dense_conv3x3_bwd_fused_supported(ldg, c0, ldx, cx) || !al16(G) || !al16(X) || !al16(GF) || !al16(Z) || !al16(scale2) ||
      !al16(shift2))
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_bwd_fused_f32: needs even ldg, c0, ldx, cx and 16-byte aligned buffers");
  const bool a16 = (ldg & 3) == 0 && (c0 & 3) == 0 && (ldx & 3) == 0 && (cx & 3) == 0;
#ifndef EML_C3_WTP   // experiment builds (tools/exp_build.sh nowtp -DEML_C3_WTP=0): round 4's weight gradient (27 tiles, z halo)
#define EML_C3_WTP 1
#endif
#if EML_C3_WTP
  const size_t lds = (size_t)(2 * kHH * kHW * kPSG + 2 * kTH * kTW * 48 + 27 * 3 * 64 + 32) * sizeof(float) + 8 * 48 * 2 * sizeof(double);   // 163 392 of 163 840 bytes
  if (a16) {
    EML_ENSURE_LDS((&conv3x3_bwd_fused_tp_kernel<true>), lds);
    hipLaunchKernelGGL(conv3x3_bwd_fused_tp_kernel<true>, dim3(grid), dim3(512), lds, (hipStream_t)stream, G, ldg, c0, W2, Z,
                       zmean, zistd, DZ, B, H, W, partials, X, ldx, cx, sB, sC, GF, scale2, shift2, partialW);
  } else {
    EML_ENSURE_LDS((&conv3x3_bwd_fused_tp_kernel<false>), lds);
    hipLaunchKernelGGL(conv3x3_bwd_fused_tp_kernel<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, G, ldg, c0, W2, Z,
                       zmean, zistd, DZ, B, H, W, partials, X, ldx, cx, sB, sC, GF, scale2, shift2, partialW);
  }
  int rc = eml::check_launch("eml_dense_conv3x3_bwd_fused_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(21 * 256 / 64), dim3(256), 0, (hipStream_t)stream, partialW, grid,
                     (size_t)21 * 256, 3, 0, 0, 0, dW2);
  return eml::check_launch("eml_dense_conv3x3_bwd_fused_f32(reduce)");
#else
  const size_t lds = (size_t)(2 * kHH * kHW * kPSG + kHH * kHW * kPSW + 27 * 3 * 64 + 32) * sizeof(float) + 8 * 48 * 2 * sizeof(double);
  if (a16) {
    EML_ENSURE_LDS((&conv3x3_bwd_fused_kernel<true>), lds);
    hipLaunchKernelGGL(conv3x3_bwd_fused_kernel<true>, dim3(grid), dim3(512), lds, (hipStream_t)stream, G, ldg, c0, W2, Z, zmean,
                       zistd, DZ, B, H, W, partials, X, ldx, cx, sB, sC, GF, scale2, shift2, partialW);
  } else {
    EML_ENSURE_LDS((&conv3x3_bwd_fused_kernel<false>), lds);
    hipLaunchKernelGGL(conv3x3_bwd_fused_kernel<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, G, ldg, c0, W2, Z, zmean,
                       zistd, DZ, B, H, W, partials, X, ldx, cx, sB, sC, GF, scale2, shift2, partialW);
  }
  int rc = eml::check_launch("eml_dense_conv3x3_bwd_fused_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(27 * 256 / 64), dim3(256), 0, (hipStream_t)stream, partialW, 2 * grid,
                     (size_t)27 * 256, 1, 0, 0, 0, dW2);
  return eml::check_launch("eml_dense_conv3x3_bwd_fused_f32(reduce)");
#endif
}

extern "C" int eml_dense_bn_bwd_finalize_f32(const double* partials, int R, int pstride, double count,
                                             const float* gamma, const float* mean, const float* istd, int C, int Cpad,
                                             int training, float* dgamma, float* dbeta, float* cA, float* cB,
                                             float* cC, float* sB, float* sC, int s_accumulate, int c_lo,
                                             int c_hi, const float* beta, const float* W, const float* dW, int w_rows,
                                             int* cond, int* any_cond, eml_stream_t stream) {
  if (!partials || !gamma || !mean || !istd || !dgamma || !dbeta || C < 1 || Cpad < C || R < 1 ||
      (cA && (!cB || !cC)) || (sB && !sC) || c_lo < 0 || c_hi <= c_lo)
    return eml::fail(EML_EINVAL, "eml_dense_bn_bwd_finalize_f32: bad arguments");
  if (W && (!dW || !beta || w_rows < 1))
    return eml::fail(EML_EINVAL, "eml_dense_bn_bwd_finalize_f32: W given without dW / beta / w_rows");
  if (c_hi > Cpad) c_hi = Cpad;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c_hi - c_lo + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials,
                     R, pstride, count, gamma, mean, istd, C, Cpad, training, dgamma, dbeta, cA, cB, cC, sB, sC,
                     s_accumulate, c_lo, c_hi, beta, W, dW, w_rows, W ? cond : nullptr, W ? any_cond : nullptr);
  return eml::check_launch("eml_dense_bn_bwd_finalize_f32");
}

extern "C" int eml_dense_bn_dgamma_direct_f32(const float* X, int ldx, long P, int Hin, int Win, int pool, const float* DY,
                                              int ld_dy, const float* Zr, int ld_z, const float* cA, const float* cB,
                                              const float* cC, int Cout, const float* W, int Cin, const float* scale1,
                                              const float* shift1, const float* mean, const float* istd, const int* cond,
                                              double* scratch, float* dgamma, int grid, eml_stream_t stream) {
  if (!X || !DY || !W || !scale1 || !shift1 || !mean || !istd || !cond || !scratch || !dgamma || P < 1 || grid < 1 ||
      Cin < 1 || Cin > 384 || Cout < 1 || Cout > 192 || Cin > ldx || Cout > ld_dy || (Zr && (!cA || !cB || !cC || Cout > ld_z)) ||
      (pool && ((Hin | Win) & 1)))
    return eml::fail(EML_EINVAL, "eml_dense_bn_dgamma_direct_f32: bad arguments (Cin <= 384, Cout <= 192, even maps when pooled)");
  hipLaunchKernelGGL(bn_dgamma_direct_partial_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx, P, Hin, Win,
                     pool, DY, ld_dy, Zr, ld_z, cA, cB, cC, Cout, W, Cin, scale1, shift1, mean, istd, cond, scratch);
  int rc = eml::check_launch("eml_dense_bn_dgamma_direct_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_dgamma_direct_reduce_kernel, dim3((Cin + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch,
                     grid, Cin, cond, dgamma);
  return eml::check_launch("eml_dense_bn_dgamma_direct_f32(reduce)");
}

// partial: [grid*2][Kp][48] floats of scratch; dW: [Cout][Cin] (PyTorch layout), all chunks.
extern "C" int eml_dense_conv1x1_bwd_weight_f32(const float* X, int ldx, long P, int Hin, int Win, int pool, int Kp,
                                                int Cin, const float* scale1, const float* shift1, const float* DY,
                                                int ld_dy, const float* Zr, int ld_z, const float* cA, const float* cB,
                                                const float* cC, int Cout, float* partial, float* dW, int grid,
                                                float* dz_out, const float* W1, int k_lo, const float* G, int ldg,
                                                float* N12, double* partials_n, eml_stream_t stream) {
  if (dz_out && (Cout != 48 || pool))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_weight_f32: dz_out is for dense layers (Cout == 48, no pool)");
  if (N12 && (Cout != 48 || pool || !W1 || !G || !partials_n || k_lo < 0 || (k_lo & 1) || k_lo + 12 > Cin || (ldg & 1) ||
              (ldg != 12 && k_lo + 12 > ldg)))   // ldg == 12: G is the compact (P, 12) tensor of exactly those channels
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_weight_f32: fused narrow pass needs a dense layer, W1, G, partials_n "
                                 "and an even k_lo with k_lo + 12 <= Cin (k_lo=%d, Cin=%d)", k_lo, Cin);
  if (!X || !scale1 || !shift1 || !DY || !Zr || !cA || !cB || !cC || !partial || !dW || P < 1 || grid < 1 || Kp < 32 ||
      (Kp & 15) || Cin > Kp || Cout < 1)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_weight_f32: bad arguments");
  const int nchunks = (Cout + 47) / 48;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int n0 = ch * 48, nv = (Cout - n0 < 48) ? Cout - n0 : 48;
    int n_load = ((ld_dy < ld_z ? ld_dy : ld_z) - n0) & ~3;
    if (n_load > 48) n_load = 48;
    if (n_load < nv) return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_weight_f32: DY/Zr rows narrower than Cout");
    const NarrowArgs na{W1, G, N12, partials_n, Cin, k_lo, ldg};
    // float4 x operand: A/B only (EML_W1_VEC=4).  Measured (profiles/r04_ab_w1_vec.txt): on the layers where its 64-channel
    // groups deal evenly (Kp in [160, 256], the plain variant) 30.4 against 29.95-30.4 ms per step -- no gain, although the
    // access-shape probe streams the float4 shape 19-24 % faster: the x stream's shape is not what bounds this kernel.
    // Everywhere (uneven deals, the narrow variant spilling 39 registers): 34.6 ms.
    static const int vec_env = [] { const char* v = getenv("EML_W1_VEC"); return v ? atoi(v) : 0; }();
    const bool vec4 = !pool && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                      vec_env == 4;
    if (vec4 && N12 && vec_env == 4)   // (the narrow epilogue + float4 operand spills 39 registers: A/B only)
      hipLaunchKernelGGL((conv1x1_bwd_weight_kernel<false, true, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx,
                         (int)P, Hin, Win, Kp, scale1, shift1, DY + n0, ld_dy, Zr + n0, ld_z, cA + n0, cB + n0, cC + n0,
                         nv, n_load, partial, dz_out, na);
    else if (vec4 && !N12)
      hipLaunchKernelGGL((conv1x1_bwd_weight_kernel<false, false, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx,
                         (int)P, Hin, Win, Kp, scale1, shift1, DY + n0, ld_dy, Zr + n0, ld_z, cA + n0, cB + n0, cC + n0,
                         nv, n_load, partial, dz_out, na);
    else if (pool)
      hipLaunchKernelGGL((conv1x1_bwd_weight_kernel<true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx,
                         (int)P, Hin, Win, Kp, scale1, shift1, DY + n0, ld_dy, Zr + n0, ld_z, cA + n0, cB + n0, cC + n0,
                         nv, n_load, partial, nullptr, na);
    else if (N12)
      hipLaunchKernelGGL((conv1x1_bwd_weight_kernel<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx,
                         (int)P, Hin, Win, Kp, scale1, shift1, DY + n0, ld_dy, Zr + n0, ld_z, cA + n0, cB + n0, cC + n0,
                         nv, n_load, partial, dz_out, na);
    else
      hipLaunchKernelGGL((conv1x1_bwd_weight_kernel<false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ldx,
                         (int)P, Hin, Win, Kp, scale1, shift1, DY + n0, ld_dy, Zr + n0, ld_z, cA + n0, cB + n0, cC + n0,
                         nv, n_load, partial, dz_out, na);
    int rc = eml::check_launch("eml_dense_conv1x1_bwd_weight_f32");
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((Cin * 48 + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial,
                       grid, (size_t)Kp * 48, 0, Cin, nv, n0, dW);
    rc = eml::check_launch("eml_dense_conv1x1_bwd_weight_f32(reduce)");
    if (rc) return rc;
  }
  return EML_OK;
}

extern "C" int eml_dense_permute_w1_bwd_f32(const float* W, int Cout, int Cin, int Kp, int Ko, float* Wd,
                                            eml_stream_t stream) {
  if (!W || !Wd || Cout < 1 || Cin < 1 || Kp < Cin || (Kp & 15) || Ko < Cout || (Ko & 15))
    return eml::fail(EML_EINVAL, "eml_dense_permute_w1_bwd_f32: bad arguments");
  hipLaunchKernelGGL(permute_w1_bwd_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, W, Cout, Cin, Kp, Ko, Wd);
  return eml::check_launch("eml_dense_permute_w1_bwd_f32");
}

extern "C" int eml_dense_conv1x1_bwd_data_f32(const float* DY, int ld_dy, const float* Zr, int ld_z, const float* cA,
                                              const float* cB, const float* cC, int Ko, const float* Wd,
                                              const float* X, int ldx, const float* scale1, const float* shift1,
                                              const float* mean, const float* istd, long P, int Hin, int Win, int pool,
                                              int Kp, float* G, int ldg, int accumulate, double* partials, int grid,
                                              const unsigned short* relu_mask16, eml_stream_t stream) {
  if (!DY || !Zr || !cA || !cB || !cC || !Wd || (!relu_mask16 && (!X || !mean || !istd)) || !scale1 || !shift1 || !G ||
      !partials || P < 1 || grid < 1 || (Kp & 15) || (Ko & 15) || Ko > ld_dy || Ko > ld_z || (ld_dy & 3) || (ld_z & 3) ||
      (ldx & 3) || (ldg & 3) || (!relu_mask16 && Kp > ldx) || Kp > ldg)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_f32: bad arguments");
  if (relu_mask16 && !(pool && Ko != 48))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_f32: relu_mask16 is for the transition form (pool, Ko != 48)");
  const size_t lds = (size_t)4 * Kp * 2 * sizeof(double);
#define EML_LAUNCH_BWD_DATA(POOLV, RESV, NCHV)                                                                        \
  hipLaunchKernelGGL((conv1x1_bwd_data_kernel<POOLV, RESV, NCHV>), dim3(grid), dim3(256), lds, (hipStream_t)stream, DY, \
                     ld_dy, Zr, ld_z, cA, cB, cC, Ko, Wd, X, ldx, scale1, shift1, mean, istd, (int)P, Hin, Win, Kp, G, \
                     ldg, accumulate, partials)
  // dense layers (Ko == 48): 2 channel tiles per chunk -> fewer accumulators, 3 waves/SIMD; measured
  // 10-20 % faster than 4 (the kernel is HBM-bound: more waves = more bytes in flight)
  if (pool) {
    if (Ko == 48) {
      EML_LAUNCH_BWD_DATA(true, true, 4);
    } else {  // transitions
      const size_t lds_t = lds + (size_t)(4 * Kp + 3 * Ko + 4 * 16 * 36) * sizeof(float);   // + the staged stores' wave tiles
#define EML_LAUNCH_TRANSITION(ACCV, MSKV)                                                                              \
  do {                                                                                                                \
    EML_ENSURE_LDS((&transition_bwd_data_kernel<2, 2, ACCV, MSKV>), lds_t);                                 \
    hipLaunchKernelGGL((transition_bwd_data_kernel<2, 2, ACCV, MSKV>), dim3(grid), dim3(256), lds_t,                   \
                       (hipStream_t)stream, DY, ld_dy, Zr, ld_z, cA, cB, cC, Ko, Wd, X, ldx, scale1, shift1, mean, istd, \
                       (int)P, Hin, Win, Kp, G, ldg, partials, relu_mask16);                                           \
  } while (0)
      // EMLight's three transitions (108 / 150 / 171 output channels: Ko = 112 / 160 / 176) with the forward's ReLU bits:
      // dz fragments resident in registers for the tile (NJO = Ko / 16)
#define EML_LAUNCH_TRANSITION_RES(NJOV)                                                                                \
  do {                                                                                                                \
    EML_ENSURE_LDS((&transition_bwd_data_kernel<2, 2, false, true, NJOV>), lds_t);                                     \
    hipLaunchKernelGGL((transition_bwd_data_kernel<2, 2, false, true, NJOV>), dim3(grid), dim3(256), lds_t,            \
                       (hipStream_t)stream, DY, ld_dy, Zr, ld_z, cA, cB, cC, Ko, Wd, X, ldx, scale1, shift1, mean, istd, \
                       (int)P, Hin, Win, Kp, G, ldg, partials, relu_mask16);                                           \
  } while (0)
      if (relu_mask16 && !accumulate && Ko == 112) EML_LAUNCH_TRANSITION_RES(7);
      else if (relu_mask16 && !accumulate && Ko == 160) EML_LAUNCH_TRANSITION_RES(10);
      else if (relu_mask16 && !accumulate && Ko == 176) EML_LAUNCH_TRANSITION_RES(11);
      else if (relu_mask16) {
        if (accumulate) EML_LAUNCH_TRANSITION(true, true); else EML_LAUNCH_TRANSITION(false, true);
      } else {
        if (accumulate) EML_LAUNCH_TRANSITION(true, false); else EML_LAUNCH_TRANSITION(false, false);
      }
#undef EML_LAUNCH_TRANSITION
#undef EML_LAUNCH_TRANSITION_RES
    }
  } else {
    if (Ko == 48) EML_LAUNCH_BWD_DATA(false, true, 2); else EML_LAUNCH_BWD_DATA(false, false, 4);
  }
#undef EML_LAUNCH_BWD_DATA
  return eml::check_launch("eml_dense_conv1x1_bwd_data_f32");
}

// Dense-layer data gradient for 1 or 2 consecutive layers in one pass over channels [k_lo, k_hi).
// Arrays of length n_layers (1 or 2): DZ/Zr (P,48), cA/cB/cC (48), Wd, scale1/shift1 (Kp_j), partials
// ([grid][Kp_j][2]), Kp.  G[p][k] += sum_j scale1_j[k]*dam_j[p][k] for k in range.
namespace {
int launch_bwd_data_multi(int n_layers, const float* const* DZ, const float* const* Zr, const float* const* cA,
                          const float* const* cB, const float* const* cC, const float* const* Wd,
                          const float* const* scale1, const float* const* shift1, double* const* partials, const int* Kp,
                          const float* X, int ldx, const float* mean, const float* istd, long P, int k_lo, int k_hi,
                          float* G, int ldg, int grid, const unsigned long long* const* relu_masks, float* top,
                          eml_stream_t stream) {
  if (top && (n_layers != 2 || !relu_masks || Zr || k_lo != 0 || k_hi < 24 || (k_hi & 3) || P * 12 > 2147483647L))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_top_f32: the compact top-24 output needs two layers, "
                                 "relu_masks, materialised dz, k_lo == 0 and k_hi >= 24, a multiple of 4 (k_hi=%d)", k_hi);
  if (n_layers < 1 || n_layers > 2 || !DZ || !Wd || !scale1 || !shift1 || !partials ||
      !Kp || (!relu_masks && (!X || !mean || !istd)) || !G || P < 1 || grid < 1 || k_lo < 0 || k_hi <= k_lo ||
      (ldx & 3) || (ldg & 3))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_f32: bad arguments");
  // Zr == NULL: DZ holds the materialised dz (eml_dense_conv1x1_bwd_weight_f32's dz_out); else dz = cA*DZ + cB*Zr + cC
  const bool raw = Zr == nullptr;
  if (!raw && (!cA || !cB || !cC))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_f32: Zr given without cA/cB/cC");
  if (relu_masks && !raw)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_f32: relu_masks needs the materialised dz (Zr == NULL)");
  BwdLayer L[2];
  const unsigned long long* Mk[2] = {nullptr, nullptr};
  int kmax = 0;
  for (int j = 0; j < 2; ++j) {
    const int s = j < n_layers ? j : 0;
    const unsigned long long* mk = relu_masks ? relu_masks[s] : nullptr;
    Mk[j] = mk;
    L[j] = raw ? BwdLayer{DZ[s], nullptr, nullptr, nullptr, nullptr, Wd[s], scale1[s], shift1[s], partials[s], Kp[s]}
               : BwdLayer{DZ[s], Zr[s], cA[s], cB[s], cC[s], Wd[s], scale1[s], shift1[s], partials[s], Kp[s]};
    if (!DZ[s] || (!raw && !Zr[s]) || !Wd[s] || !partials[s] || (relu_masks && !mk) || (Kp[s] & 15) ||
        (!relu_masks && Kp[s] > ldx) || Kp[s] > ldg)
      return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_f32: bad layer %d", s);
    if (Kp[s] > kmax) kmax = Kp[s];
  }
  if (((k_hi + 15) & ~15) > kmax) return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_f32: range beyond Kp");
  const size_t lds = (size_t)n_layers * 4 * kmax * 2 * sizeof(double) + (size_t)(2 + 2 * n_layers) * kmax * sizeof(float);
  if (lds > 80 * 1024) return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_f32: Kp=%d does not fit LDS", kmax);
#define EML_LAUNCH_MULTI(NLV, MTV, RAWV, MSKV)                                                                       \
  do {                                                                                                              \
    EML_ENSURE_LDS((&conv1x1_bwd_data_multi_kernel<NLV, MTV, RAWV, MSKV>), lds);                                 \
    hipLaunchKernelGGL((conv1x1_bwd_data_multi_kernel<NLV, MTV, RAWV, MSKV>), dim3(grid), dim3(256), lds,            \
                       (hipStream_t)stream, L[0], L[1], X, ldx, mean, istd, (int)P, k_lo, k_hi, G, ldg, kmax,        \
                       Mk[0], Mk[1], (float*)nullptr);                                                              \
  } while (0)
  // two layers: 32 pixels per wave keeps both layers' dz fragments resident at 2 waves/SIMD
  if (n_layers == 1) {
    if (relu_masks) EML_LAUNCH_MULTI(1, 4, true, true);
    else if (raw) EML_LAUNCH_MULTI(1, 4, true, false);
    else EML_LAUNCH_MULTI(1, 4, false, false);
  } else if (top) {
    EML_ENSURE_LDS((&conv1x1_bwd_data_multi_kernel<2, 2, true, true, true>), lds);
    hipLaunchKernelGGL((conv1x1_bwd_data_multi_kernel<2, 2, true, true, true>), dim3(grid), dim3(256), lds,
                       (hipStream_t)stream, L[0], L[1], X, ldx, mean, istd, (int)P, k_lo, k_hi, G, ldg, kmax, Mk[0], Mk[1],
                       top);
  } else {
    if (relu_masks) EML_LAUNCH_MULTI(2, 2, true, true);
    else if (raw) EML_LAUNCH_MULTI(2, 2, true, false);
    else EML_LAUNCH_MULTI(2, 2, false, false);
  }
#undef EML_LAUNCH_MULTI
  return eml::check_launch("eml_dense_conv1x1_bwd_data_multi_f32");
}
}  // namespace

extern "C" int eml_dense_conv1x1_bwd_data_multi_f32(int n_layers, const float* const* DZ, const float* const* Zr,
                                                    const float* const* cA, const float* const* cB,
                                                    const float* const* cC, const float* const* Wd,
                                                    const float* const* scale1, const float* const* shift1,
                                                    double* const* partials, const int* Kp, const float* X, int ldx,
                                                    const float* mean, const float* istd, long P, int k_lo, int k_hi,
                                                    float* G, int ldg, int grid,
                                                    const unsigned long long* const* relu_masks, eml_stream_t stream) {
  return launch_bwd_data_multi(n_layers, DZ, Zr, cA, cB, cC, Wd, scale1, shift1, partials, Kp, X, ldx, mean, istd, P, k_lo,
                               k_hi, G, ldg, grid, relu_masks, nullptr, stream);
}

// The two-layer masked pass with its top 24 channels [k_hi - 24, k_hi) written to `top` (2, P, 12) instead of G: top[0] =
// channels [k_hi - 24, k_hi - 12), top[1] = [k_hi - 12, k_hi) -- the finished-but-for-their-own-pair gradients the next
// pair of layers reads (eml_dense_conv3x3_bwd_*'s G operand with ldg = 12, c0 = 0; eml_dense_conv1x1_bwd_weight_f32's
// narrow G operand with ldg = 12).  G keeps its old values in those 24 columns.
extern "C" int eml_dense_conv1x1_bwd_data_multi_top_f32(const float* const* DZ, const float* const* Wd,
                                                        const float* const* scale1, const float* const* shift1,
                                                        double* const* partials, const int* Kp, long P, int k_hi,
                                                        float* G, int ldg, int grid,
                                                        const unsigned long long* const* relu_masks, float* top,
                                                        eml_stream_t stream) {
  if (!top) return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_data_multi_top_f32: top is NULL");
  return launch_bwd_data_multi(2, DZ, nullptr, nullptr, nullptr, nullptr, Wd, scale1, shift1, partials, Kp, nullptr, 0,
                               nullptr, nullptr, P, 0, k_hi, G, ldg, grid, relu_masks, top, stream);
}

// Narrow pass: N12 (P,12) = G[:, k_lo:k_lo+12] + scale1 * relu-mask * (dz W1[:, k_lo:k_lo+12]) and the BN1 partial sums
// of those 12 channels into partials[grid][Kp][2] (only entries k_lo..k_lo+11 are written).
extern "C" int eml_dense_conv1x1_bwd_narrow_f32(const float* DZ, const float* W1, int Cin, int k_lo, const float* X,
                                                int ldx, const float* scale1, const float* shift1, const float* mean,
                                                const float* istd, long P, const float* G, int ldg, float* N12,
                                                double* partials, int Kp, int grid, eml_stream_t stream) {
  if (!DZ || !W1 || !X || !scale1 || !shift1 || !mean || !istd || !G || !N12 || !partials || P < 1 || grid < 1 ||
      k_lo < 0 || (k_lo & 1) || k_lo + 12 > Cin || Cin > Kp || (ldx & 1) || (ldg & 1) || Kp > ldx || Kp > ldg)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_narrow_f32: bad arguments");
  hipLaunchKernelGGL((conv1x1_bwd_narrow_kernel<4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, DZ, W1, Cin, k_lo, X,
                     ldx, scale1, shift1, mean, istd, (int)P, G, ldg, N12, partials, Kp);
  return eml::check_launch("eml_dense_conv1x1_bwd_narrow_f32");
}

extern "C" int eml_dense_grad_materialize_f32(float* G, int ldg, const float* X, int ldx, const float* sB,
                                              const float* sC, int c0, int n, long P, eml_stream_t stream) {
  if (!G || !X || !sB || !sC || P < 1 || n < 2 || (n & 1) || (c0 & 1) || (ldg & 1) || (ldx & 1))
    return eml::fail(EML_EINVAL, "eml_dense_grad_materialize_f32: bad arguments");
  const size_t total = (size_t)P * (n >> 1);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(grad_materialize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, ldg, X, ldx, sB, sC, c0,
                     n, (size_t)P);
  return eml::check_launch("eml_dense_grad_materialize_f32");
}

extern "C" int eml_dense_bn_bwd_stats_f32(const float* DY, int ld_dy, const float* raw, int ld_raw, const float* out,
                                          int ld_out, int relu, int C, long P, const float* mean, const float* istd,
                                          double* partials, int grid, eml_stream_t stream) {
  if (!DY || !raw || !mean || !istd || !partials || (relu && !out) || C < 1 || C > 384 || P < 1 || grid < 1)
    return eml::fail(EML_EINVAL, "eml_dense_bn_bwd_stats_f32: bad arguments");
  auto a16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
  if (C <= 32 && (C & 3) == 0 && ((ld_dy | ld_raw | (relu ? ld_out : 0)) & 3) == 0 && a16(DY) && a16(raw) && (!relu || a16(out)) &&
      a16(mean) && a16(istd)) {
    hipLaunchKernelGGL(bn_bwd_stats_small_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, DY, ld_dy, raw, ld_raw, out, ld_out,
                       relu, C, (size_t)P, mean, istd, partials);
    return eml::check_launch("eml_dense_bn_bwd_stats_f32");
  }
  const size_t lds = (size_t)4 * ((C + 63) / 64) * 64 * 2 * sizeof(double);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, DY, ld_dy, raw, ld_raw, out,
                     ld_out, relu, C, (size_t)P, mean, istd, partials);
  return eml::check_launch("eml_dense_bn_bwd_stats_f32");
}

extern "C" int eml_dense_conv0_bwd_weight_f32(const float* x, const float* G, int ldg, const float* X1, int ldx,
                                              const float* Y0, int C0, const float* cA, const float* cB,
                                              const float* cC, int B, int H, int W, float* partial, float* dW0,
                                              int grid, eml_stream_t stream) {
  if (!x || !G || !X1 || !Y0 || !cA || !cB || !cC || !partial || !dW0 || C0 < 1 || C0 > 32 || B < 1 || grid < 1)
    return eml::fail(EML_EINVAL, "eml_dense_conv0_bwd_weight_f32: bad arguments");
  hipLaunchKernelGGL(conv0_bwd_weight_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, G, ldg, X1, ldx, Y0, C0,
                     cA, cB, cC, B, H, W, partial, nullptr, nullptr, nullptr, nullptr);
  int rc = eml::check_launch("eml_dense_conv0_bwd_weight_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(27 * 32 / 64 + 1), dim3(256), 0, (hipStream_t)stream, partial, grid * 4,
                     (size_t)1024, 2, 0, C0, 0, dW0);
  return eml::check_launch("eml_dense_conv0_bwd_weight_f32(reduce)");
}

extern "C" int eml_dense_head_pool_bwd_f32(const float* gpooled, const float* F, int ldf, int C, int B, int H, int W,
                                           int k, float* dF, int ldd, eml_stream_t stream) {
  if (!gpooled || !F || !dF || C < 1 || B < 1 || k < 1) return eml::fail(EML_EINVAL, "eml_dense_head_pool_bwd_f32: bad arguments");
  const size_t total = (size_t)B * H * W * C;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(head_pool_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gpooled, F, ldf, C, B, H, W, k,
                     dF, ldd);
  return eml::check_launch("eml_dense_head_pool_bwd_f32");
}

// Block 1's input channels without the grad_materialize pass and without the block buffer (round 6): the statistics of
// norm0 + relu0's backward and conv0's weight gradient take G as the data-gradient passes left it (the deferred BN1 affine
// sB, sC still outstanding) and the raw conv0 output Y0; x = relu(scale0*Y0 + shift0) is what eml_dense_bn_apply_f32 wrote
// into the block buffer (same expression), g = G + sB*x + sC what eml_dense_grad_materialize_f32 would have written.
extern "C" int eml_dense_norm0_bwd_stats_f32(const float* G, int ldg, const float* Y0, int ld_y0, const float* scale0,
                                             const float* shift0, const float* sB, const float* sC, int C, long P,
                                             const float* mean, const float* istd, double* partials, int grid,
                                             eml_stream_t stream) {
  auto a16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
  if (!G || !Y0 || !scale0 || !shift0 || !sB || !sC || !mean || !istd || !partials || C < 4 || C > 32 || (C & 3) || P < 1 ||
      grid < 1 || ((ldg | ld_y0) & 3) || !a16(G) || !a16(Y0) || !a16(scale0) || !a16(shift0) || !a16(sB) || !a16(sC) || !a16(mean) ||
      !a16(istd))
    return eml::fail(EML_EINVAL, "eml_dense_norm0_bwd_stats_f32: bad arguments (C a multiple of 4 <= 32, 16-byte aligned operands)");
  hipLaunchKernelGGL(norm0_bwd_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, ldg, Y0, ld_y0, scale0, shift0, sB,
                     sC, C, (size_t)P, mean, istd, partials);
  return eml::check_launch("eml_dense_norm0_bwd_stats_f32");
}

extern "C" int eml_dense_conv0_bwd_weight_fused_f32(const float* x, const float* G, int ldg, const float* Y0, int C0,
                                                    const float* scale0, const float* shift0, const float* sB, const float* sC,
                                                    const float* cA, const float* cB, const float* cC, int B, int H, int W,
                                                    float* partial, float* dW0, int grid, eml_stream_t stream) {
  if (!x || !G || !Y0 || !scale0 || !shift0 || !sB || !sC || !cA || !cB || !cC || !partial || !dW0 || C0 < 1 || C0 > 32 || B < 1 ||
      grid < 1)
    return eml::fail(EML_EINVAL, "eml_dense_conv0_bwd_weight_fused_f32: bad arguments");
  hipLaunchKernelGGL(conv0_bwd_weight_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, G, ldg, nullptr, 0, Y0, C0, cA,
                     cB, cC, B, H, W, partial, scale0, shift0, sB, sC);
  int rc = eml::check_launch("eml_dense_conv0_bwd_weight_fused_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(27 * 32 / 64 + 1), dim3(256), 0, (hipStream_t)stream, partial, grid * 4,
                     (size_t)1024, 2, 0, C0, 0, dW0);
  return eml::check_launch("eml_dense_conv0_bwd_weight_fused_f32(reduce)");
}
