// DenseNet-BC encoder, backward of a PAIR of dense layers: both 1x1 weight gradients and the two-layer data gradient in ONE
// pass over the block buffer (reference: RegressionNetwork/DenseNet.py:26-55, the autograd backward of conv1 / norm1 of two
// consecutive _DenseLayers).
//
// Round 4's pair (a = upper layer with k + 12 inputs, b = lower layer with k) moved, per pixel:
//     weight gradient of a (+ the narrow pass riding on it)   x[:k+12] + dzn_a + z_a in, dz_a out, G slice / N12        k + 180
//     weight gradient of b                                     x[:k]    + dzn_b + z_b in, dz_b out                       k + 144
//     data gradient of both                                    dz_a + dz_b in, G[:k] read-modify-write (+ mask bits)    2k +  96
// = 4k + 420 floats.  The three passes re-read what they share -- x twice, each dz once more -- because the 3x3 backward of
// layer b (a halo and a batch reduction: grid-wide dependencies) sits between a's weight-gradient pass (whose narrow data pass
// produces b's finished output gradient) and everything that needs dz_b.  Here the narrow pass runs on its own again (it
// rebuilds dz_a from dzn_a / z_a and materialises it: 180 floats per pixel, conv1x1_bwd_narrow2_kernel), layer a's weight
// gradient is POSTPONED, and one kernel then does the rest:  x[:k+12] once, dz_a, dzn_b + z_b, G[:k] read-modify-write
// = 3k + 156: **3k + 336 per pair, -(k + 84), -22 %**.
//
// conv1x1_bwd_pair_kernel: persistent workgroups over 64-pixel chunks; both layers' dz tiles of a chunk are built once in LDS
// (as the weight-gradient kernel does for one).  The channel axis is dealt to the four waves in 32-channel groups
// (wave w owns groups w, w + 4, w + 8); for each of its groups and each chunk a wave runs
//   (1) the WEIGHT-GRADIENT part, MFMA k = pixels: x as float2 per (pixel quad, group) -- lane i carries channels 2i, 2i + 1,
//       rows of two accumulator tiles -- through each layer's own BN1 affine + ReLU, against that layer's dz fragments from
//       LDS: 2 x 2 x 3 accumulator tiles per group, persistent over the chunks;
//   (2) the DATA-GRADIENT part, MFMA k = the 48 bottleneck channels: per 16-pixel tile, (W1 fragments from L2) x (dz fragments
//       from LDS) -> 4 channels of one pixel per lane; ReLU mask from the bits the forward stored (64-bit words at wave-uniform
//       addresses: scalar loads, the select masks of v_cndmask), scaled by BN1's gamma * istd and added to the G tile held in
//       registers for BOTH layers, then stored;
//       BN1's S1 = sum of the masked gradient per channel into wave-private f64 accumulators in LDS.
// No inter-wave traffic except the dz tiles; one barrier pair per chunk.  All reductions are per-workgroup partials + a
// fixed-order finishing kernel: deterministic, no atomics.
#include "eml_common.h"

#include <cstdint>
#include <cstdlib>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

struct PairLayer {
  const float* DZ;       // (P,48): dzn (with Zr / cA / cB / cC) or the materialised dz (Zr == NULL)
  const float* Zr;       // (P,48) raw conv1 output, or NULL
  const float* cA;
  const float* cB;
  const float* cC;       // dz = cA*DZ + cB*Zr + cC
  const float* Wd;       // [Kp/16][3][4][16][4] (eml_dense_permute_w1_bwd_f32)
  const float* scale1;
  const float* shift1;   // BN1 affine (Kp, zero-padded)
  double* stats;         // [grid][Kp][2]: S1 of the data gradient (slot 1 = 0: S2 comes from the weight gradient)
  float* wpartial;       // [grid][Kp][48] weight-gradient partials
  const unsigned long long* mask;   // the forward's ReLU bits: word ((p / 16) * (Kp / 16) + nt) * 4 + g, bit = lane (r, kk)
  int Kp, Cin;
};

constexpr int kMaxG = 3;   // channel groups per wave: Kp <= 384

// dz tile of one layer for a 64-pixel chunk -> LDS [64][48]; pixel = tid >> 2, columns 4 * ((tid & 3) + 4 j)
__device__ __forceinline__ void stage_dz(const PairLayer& L, int chunk, int P, float* __restrict__ dst) {
  const int spix = threadIdx.x >> 2, sq = threadIdx.x & 3;
  const int p = chunk * 64 + spix;
  const bool pv = p < P;
  const size_t pc = pv ? p : 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = 4 * (sq + 4 * j);
    float4 v = *reinterpret_cast<const float4*>(L.DZ + pc * 48 + col);
    if (L.Zr) {   // workgroup-uniform
      const float4 z = *reinterpret_cast<const float4*>(L.Zr + pc * 48 + col);
      const float4 a = *reinterpret_cast<const float4*>(L.cA + col), b = *reinterpret_cast<const float4*>(L.cB + col);
      const float4 c = *reinterpret_cast<const float4*>(L.cC + col);
      v.x = fmaf(a.x, v.x, fmaf(b.x, z.x, c.x));
      v.y = fmaf(a.y, v.y, fmaf(b.y, z.y, c.y));
      v.z = fmaf(a.z, v.z, fmaf(b.z, z.z, c.z));
      v.w = fmaf(a.w, v.w, fmaf(b.w, z.w, c.w));
    }
    if (!pv) v = make_float4(0.f, 0.f, 0.f, 0.f);   // pixels past P add nothing to either product
    *reinterpret_cast<float4*>(dst + spix * 48 + col) = v;
  }
}

__global__ __launch_bounds__(256, 1) void conv1x1_bwd_pair_kernel(PairLayer La, PairLayer Lb, const float* __restrict__ X,
                                                                  int ldx, int P, int k_hi /* data gradient: channels [0, k_hi) */,
                                                                  float* __restrict__ Gd, int ldg, int KpMax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dz_l = smem;                                              // [2 layers][64 * 48]
  double* sacc = reinterpret_cast<double*>(smem + 2 * 64 * 48);    // [2 layers][KpMax]: S1, channel-owned by one wave
  float* vec_l = reinterpret_cast<float*>(sacc + 2 * KpMax);       // [2 layers][scale1 | shift1][KpR]: BN1 affines, zero-padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const PairLayer Ls[2] = {La, Lb};
  const int ngroups = (KpMax + 31) >> 5;
  const int KpR = 32 * ngroups;                                    // whole groups: relu(0 * x + 0) = 0 for the padding channels
  for (int e = tid; e < 2 * KpMax; e += 256) sacc[e] = 0.0;
  for (int e = tid; e < KpR; e += 256) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      vec_l[(2 * j) * KpR + e] = e < Ls[j].Kp ? Ls[j].scale1[e] : 0.f;
      vec_l[(2 * j + 1) * KpR + e] = e < Ls[j].Kp ? Ls[j].shift1[e] : 0.f;
    }
  }
  const int ngw = ngroups > wave ? (ngroups - wave + 3) >> 2 : 0;   // groups of this wave (wave-uniform)
  f32x4 acc[kMaxG][2][2][3];   // [group][layer][t][o tile]
#pragma unroll
  for (int i = 0; i < kMaxG; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[i][j][t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = (P + 63) >> 6;
  // Software pipeline (one wave per SIMD: nothing else hides a load's latency).  The weight-gradient operand of a group -- 16
  // float2, one per pixel quad -- is requested during the PREVIOUS group's data-gradient part (across chunk boundaries too);
  // the data-gradient operands of a group (x and G tiles of its four pixel tiles, the W1 fragments) are requested at the top
  // of its weight-gradient part and arrive behind its 192 MFMAs.
  float2 xw[16];
  auto issue_xw = [&](int ch, int cg) {
    const int cw = min(32 * cg + 2 * r, ldx - 2);
    const int chc = min(ch, nchunks - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pc = min(chc * 64 + 4 * q + kk, P - 1);
      xw[q] = *reinterpret_cast<const float2*>(X + (size_t)pc * ldx + cw);
    }
  };
  if (ngw > 0) issue_xw(blockIdx.x, wave);
  __syncthreads();
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    stage_dz(La, chunk, P, dz_l);
    stage_dz(Lb, chunk, P, dz_l + 64 * 48);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kMaxG; ++i) {
      if (i >= ngw) break;   // wave-uniform
      const int cg = wave + 4 * i;
      const bool dgrad = 32 * cg < k_hi;   // wave-uniform: else the group lies in layer a's own 12 new inputs / the padding
      // ---- data-gradient operands of this group: requested now, used after the weight-gradient MFMAs
      float4 wq[2][2][3], gq[4][2];
      if (dgrad) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nnt = Ls[j].Kp >> 4;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int nt = min(2 * cg + n, nnt - 1);
#pragma unroll
            for (int jo = 0; jo < 3; ++jo)
              wq[j][n][jo] = *reinterpret_cast<const float4*>(Ls[j].Wd + ((((size_t)nt * 3 + jo) * 4 + kk) * 16 + r) * 4);
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const size_t prow = (size_t)min(chunk * 64 + 16 * m + r, P - 1);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int k4 = 32 * cg + 16 * n + 4 * kk;
            gq[m][n] = *reinterpret_cast<const float4*>(Gd + prow * ldg + (k4 < k_hi ? k4 : 0));
          }
        }
      }
      // ---------------------------------------------------------------- (1) weight gradients of both layers
      {
        float2 sw[2], tw[2];   // this lane's two channels 32cg + 2r + t, per layer
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          sw[j] = *reinterpret_cast<const float2*>(vec_l + (2 * j) * KpR + 32 * cg + 2 * r);
          tw[j] = *reinterpret_cast<const float2*>(vec_l + (2 * j + 1) * KpR + 32 * cg + 2 * r);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int pl = 4 * q + kk;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float* dzb = dz_l + j * 64 * 48 + pl * 48 + r;
            const float bz0 = dzb[0], bz1 = dzb[16], bz2 = dzb[32];
            const float a0 = fmaxf(fmaf(xw[q].x, sw[j].x, tw[j].x), 0.f);
            const float a1 = fmaxf(fmaf(xw[q].y, sw[j].y, tw[j].y), 0.f);
            acc[i][j][0][0] = mfma16(a0, bz0, acc[i][j][0][0]);
            acc[i][j][1][0] = mfma16(a1, bz0, acc[i][j][1][0]);
            acc[i][j][0][1] = mfma16(a0, bz1, acc[i][j][0][1]);
            acc[i][j][1][1] = mfma16(a1, bz1, acc[i][j][1][1]);
            acc[i][j][0][2] = mfma16(a0, bz2, acc[i][j][0][2]);
            acc[i][j][1][2] = mfma16(a1, bz2, acc[i][j][1][2]);
          }
          // keep the scheduler from hoisting every quad's LDS reads to the top (96 live registers: it spilled)
          if (q & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next group's weight-gradient operand (this chunk's next group of the wave, or the first one of its next chunk)
      {
        const bool more = i + 1 < ngw;
        issue_xw(more ? chunk : chunk + (int)gridDim.x, more ? cg + 4 : wave);
      }
      if (!dgrad) continue;
      // ---------------------------------------------------------------- (2) data gradient of both layers, channels < k_hi
      float4 sk[2][2];   // BN1's gamma * istd of this lane's channels 32cg + 16n + 4kk .. + 3, per layer
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n) sk[j][n] = *reinterpret_cast<const float4*>(vec_l + (2 * j) * KpR + 32 * cg + 16 * n + 4 * kk);
      // which of this lane's channels lie in [0, k_hi), as lane masks (ANDed into the ReLU words on the scalar unit)
      unsigned long long inb[2][4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) inb[n][g] = __ballot(32 * cg + 16 * n + 4 * kk + g < k_hi);
      float l1[2][2][4];   // S1 of this chunk: [layer][channel tile][channel]
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) l1[j][n][g] = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int p = chunk * 64 + 16 * m + r;
        const bool pv = p < P;
        const size_t prow = (size_t)min(p, P - 1);
        const unsigned long long pvb = __ballot(pv);
        // the forward's ReLU bits of this pixel tile's (channel tile, channel) words: wave-uniform addresses -> scalar loads
        unsigned long long mk[2][2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nnt = Ls[j].Kp >> 4;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const size_t wi = ((size_t)__builtin_amdgcn_readfirstlane(chunk * 4 + m) * nnt + min(2 * cg + n, nnt - 1)) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) mk[j][n][g] = Ls[j].mask[wi + g];
          }
        }
        // dz fragments of this pixel tile: lane (r, kk) takes dz[p = 16m + r][16jo + 4kk .. + 3]
        float4 dzf[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int jo = 0; jo < 3; ++jo)
            dzf[j][jo] = *reinterpret_cast<const float4*>(dz_l + j * 64 * 48 + (16 * m + r) * 48 + 16 * jo + 4 * kk);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 da[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int jo = 0; jo < 3; ++jo)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int n = 0; n < 2; ++n) da[n] = mfma16(f4c(wq[j][n][jo], t), f4c(dzf[j][jo], t), da[n]);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const float4 s = sk[j][n];
            // bit `lane` of word g = this lane's ReLU mask for channel g of its quad: the word IS v_cndmask's select mask
            const float d0 = __builtin_amdgcn_inverse_ballot_w64(mk[j][n][0] & pvb & inb[n][0]) ? da[n][0] : 0.f;
            const float d1 = __builtin_amdgcn_inverse_ballot_w64(mk[j][n][1] & pvb & inb[n][1]) ? da[n][1] : 0.f;
            const float d2 = __builtin_amdgcn_inverse_ballot_w64(mk[j][n][2] & pvb & inb[n][2]) ? da[n][2] : 0.f;
            const float d3 = __builtin_amdgcn_inverse_ballot_w64(mk[j][n][3] & pvb & inb[n][3]) ? da[n][3] : 0.f;
            gq[m][n].x = fmaf(s.x, d0, gq[m][n].x);
            gq[m][n].y = fmaf(s.y, d1, gq[m][n].y);
            gq[m][n].z = fmaf(s.z, d2, gq[m][n].z);
            gq[m][n].w = fmaf(s.w, d3, gq[m][n].w);
            l1[j][n][0] += d0;
            l1[j][n][1] += d1;
            l1[j][n][2] += d2;
            l1[j][n][3] += d3;
          }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int k4 = 32 * cg + 16 * n + 4 * kk;
          if (pv && k4 < k_hi) *reinterpret_cast<float4*>(Gd + prow * ldg + k4) = gq[m][n];
        }
        __builtin_amdgcn_sched_barrier(0);   // one pixel tile's fragments and mask words at a time
      }
      // S1 of the chunk: over the 16 pixel lanes (DPP), then into this wave's own channels of the f64 accumulators
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float tot = eml::row16_sum(l1[j][n][g]);
            const int c = 32 * cg + 16 * n + 4 * kk + g;
            if (r == 0 && c < KpMax) sacc[j * KpMax + c] += (double)tot;
          }
    }
    __syncthreads();   // every wave is done with this chunk's dz tiles
  }
  // ---- weight-gradient partials: D element g of tile (t, n): row 4kk + g <-> channel 32cg + 2(4kk + g) + t, column r <-> o
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float* out = Ls[j].wpartial + (size_t)blockIdx.x * Ls[j].Kp * 48;
#pragma unroll
    for (int i = 0; i < kMaxG; ++i) {
      const int cg = wave + 4 * i;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = 32 * cg + 2 * (4 * kk + g) + t;
          if (cg < ngroups && ch < Ls[j].Kp) {
#pragma unroll
            for (int n = 0; n < 3; ++n) out[(size_t)ch * 48 + 16 * n + r] = acc[i][j][t][n][g];
          }
        }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int Kp = Ls[j].Kp;
    for (int e = tid; e < Kp; e += 256) {
      Ls[j].stats[((size_t)blockIdx.x * Kp + e) * 2] = sacc[j * KpMax + e];
      Ls[j].stats[((size_t)blockIdx.x * Kp + e) * 2 + 1] = 0.0;
    }
  }
}

// dW[o][k] = sum over the workgroups' partials [R][Kp][48] (f64, fixed order)
__global__ __launch_bounds__(256) void pair_reduce_kernel(const float* __restrict__ partial, int R, int Kp, int Cin,
                                                          float* __restrict__ dW) {
  __shared__ double red[4][64];
  const int tid = threadIdx.x, e = blockIdx.x * 64 + (tid & 63), slice = tid >> 6;
  const int k = e / 48, o = e - 48 * k;
  const bool valid = k < Cin;
  double s = 0.0;
  if (valid) {
#pragma unroll 8
    for (int rr = slice; rr < R; rr += 4) s += (double)partial[(size_t)rr * Kp * 48 + e];
  }
  red[slice][tid & 63] = s;
  __syncthreads();
  if (slice == 0 && valid) dW[(size_t)o * Cin + k] = (float)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
}

// The narrow pass on its own, from the UN-materialised operands: dz = cA*DZ + cB*Zr + cC is rebuilt in registers, written to
// dz_out (P,48; may alias DZ: every element is read and written by the same thread), and
// N12[p][c] = G[p][k_lo+c] + scale1[k_lo+c] * mask * sum_o dz[p][o] W1[o][k_lo+c]; (S1, S2) = (sum dam, sum dam * xhat) of the
// 12 channels -> partials, BOTH accumulated here per element in f64: the lower layer's 3x3 backward needs this range's BN1
// backward before the (postponed) weight gradient it would otherwise be derived from exists.
template <int MT>
__global__ __launch_bounds__(256, 2) void conv1x1_bwd_narrow2_kernel(
    const float* __restrict__ DZ, const float* __restrict__ Zr, const float* __restrict__ cA, const float* __restrict__ cB,
    const float* __restrict__ cC, float* __restrict__ dz_out, const float* __restrict__ W1 /*[48][Cin]*/, int Cin, int k_lo,
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
  float4 ca[3], cb[3], cc[3];
#pragma unroll
  for (int jo = 0; jo < 3; ++jo) {
    ca[jo] = *reinterpret_cast<const float4*>(cA + 16 * jo + 4 * kk);
    cb[jo] = *reinterpret_cast<const float4*>(cB + 16 * jo + 4 * kk);
    cc[jo] = *reinterpret_cast<const float4*>(cC + 16 * jo + 4 * kk);
  }
  const bool cv = kk < 3;                   // lanes of row group 3 hold the padding channels 12..15
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
      for (int jo = 0; jo < 3; ++jo) {
        const float4 d = *reinterpret_cast<const float4*>(DZ + prow[m] * 48 + 16 * jo + 4 * kk);
        const float4 z = *reinterpret_cast<const float4*>(Zr + prow[m] * 48 + 16 * jo + 4 * kk);
        dz[m][jo].x = fmaf(ca[jo].x, d.x, fmaf(cb[jo].x, z.x, cc[jo].x));
        dz[m][jo].y = fmaf(ca[jo].y, d.y, fmaf(cb[jo].y, z.y, cc[jo].y));
        dz[m][jo].z = fmaf(ca[jo].z, d.z, fmaf(cb[jo].z, z.z, cc[jo].z));
        dz[m][jo].w = fmaf(ca[jo].w, d.w, fmaf(cb[jo].w, z.w, cc[jo].w));
      }
      xa[m] = *reinterpret_cast<const float2*>(X + prow[m] * ldx + cq);       // k_lo is even: 8-byte aligned
      xb[m] = *reinterpret_cast<const float2*>(X + prow[m] * ldx + cq + 2);
      ga[m] = *reinterpret_cast<const float2*>(Gd + prow[m] * ldg + cq);
      gb[m] = *reinterpret_cast<const float2*>(Gd + prow[m] * ldg + cq + 2);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (pv[m]) {
#pragma unroll
        for (int jo = 0; jo < 3; ++jo) *reinterpret_cast<float4*>(dz_out + prow[m] * 48 + 16 * jo + 4 * kk) = dz[m][jo];
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

}  // namespace

extern "C" int eml_dense_conv1x1_bwd_narrow2_f32(const float* DZ, const float* Zr, const float* cA, const float* cB,
                                                 const float* cC, float* dz_out, const float* W1, int Cin, int k_lo,
                                                 const float* X, int ldx, const float* scale1, const float* shift1,
                                                 const float* mean, const float* istd, long P, const float* G, int ldg,
                                                 float* N12, double* partials, int Kp, int grid, eml_stream_t stream) {
  if (!DZ || !Zr || !cA || !cB || !cC || !dz_out || !W1 || !X || !scale1 || !shift1 || !mean || !istd || !G || !N12 || !partials || P < 1 ||
      grid < 1 || k_lo < 0 || (k_lo & 1) || k_lo + 12 > Cin || Cin > Kp || (ldx & 1) || (ldg & 1) || Kp > ldx || Kp > ldg)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_narrow2_f32: bad arguments (k_lo=%d, Cin=%d, Kp=%d)", k_lo, Cin, Kp);
  hipLaunchKernelGGL((conv1x1_bwd_narrow2_kernel<4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, DZ, Zr, cA, cB, cC, dz_out,
                     W1, Cin, k_lo, X, ldx, scale1, shift1, mean, istd, (int)P, G, ldg, N12, partials, Kp);
  return eml::check_launch("eml_dense_conv1x1_bwd_narrow2_f32");
}

// Both 1x1 weight gradients and the data gradient of a pair of dense layers in one pass (see the head of this file).
// Arrays of length 2, [0] = the UPPER layer a (Cin_a = Cin_b + 12), [1] = the lower layer b:
//   DZ / Zr / cA / cB / cC   dz = cA*DZ + cB*Zr + cC, or Zr[j] == NULL: DZ[j] is the materialised dz
//   Wd                       eml_dense_permute_w1_bwd_f32's layout (Kp_j x 48)
//   scale1 / shift1          BN1 affine (Kp_j floats, zero-padded)
//   stats                    [grid][Kp_j][2] f64 out: S1 of the data gradient over channels [0, Cin_b)
//   wpartial / dW            [grid][Kp_j][48] scratch and the finished (48, Cin_j) weight gradient
//   relu_masks               the forward's ReLU bits of BN1's output (eml_dense_conv1x1_fwd_f32's relu_mask), per layer
// G[:, 0:Cin_b) += scale1_a * dam_a + scale1_b * dam_b.
extern "C" int eml_dense_conv1x1_bwd_pair_f32(const float* const* DZ, const float* const* Zr, const float* const* cA,
                                              const float* const* cB, const float* const* cC, const float* const* Wd,
                                              const float* const* scale1, const float* const* shift1, double* const* stats,
                                              float* const* wpartial, float* const* dW, const int* Kp, const int* Cin,
                                              const unsigned long long* const* relu_masks, const float* X, int ldx, long P,
                                              float* G, int ldg, int grid, eml_stream_t stream) {
  if (!DZ || !Zr || !cA || !cB || !cC || !Wd || !scale1 || !shift1 || !stats || !wpartial || !dW || !Kp || !Cin || !relu_masks || !X || !G ||
      P < 1 || grid < 1 || (ldx & 3) || (ldg & 3))
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_pair_f32: null argument / bad strides");
  PairLayer L[2];
  for (int j = 0; j < 2; ++j) {
    if (!DZ[j] || !Wd[j] || !scale1[j] || !shift1[j] || !stats[j] || !wpartial[j] || !dW[j] || !relu_masks[j] ||
        (Zr[j] && (!cA[j] || !cB[j] || !cC[j])) ||
        Kp[j] < 32 || (Kp[j] & 15) || Kp[j] > 32 * 4 * kMaxG || Cin[j] < 1 || Cin[j] > Kp[j] || Kp[j] > ldx || Kp[j] > ldg)
      return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_pair_f32: bad layer %d (Kp=%d, Cin=%d)", j, Kp[j], Cin[j]);
    L[j] = PairLayer{DZ[j], Zr[j], cA[j], cB[j], cC[j], Wd[j], scale1[j], shift1[j], stats[j], wpartial[j], relu_masks[j], Kp[j], Cin[j]};
  }
  if (Cin[0] != Cin[1] + 12)
    return eml::fail(EML_EINVAL, "eml_dense_conv1x1_bwd_pair_f32: the upper layer must have the lower one's inputs + 12 "
                                 "(Cin = %d, %d)", Cin[0], Cin[1]);
  const int kmax = Kp[0] > Kp[1] ? Kp[0] : Kp[1];
  const size_t lds = (size_t)2 * 64 * 48 * sizeof(float) + (size_t)2 * kmax * sizeof(double) +
                     (size_t)4 * 32 * ((kmax + 31) / 32) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  EML_ENSURE_LDS((&conv1x1_bwd_pair_kernel), lds);
  hipLaunchKernelGGL(conv1x1_bwd_pair_kernel, dim3(grid), dim3(256), lds, st, L[0], L[1], X, ldx, (int)P, Cin[1], G, ldg, kmax);
  int rc = eml::check_launch("eml_dense_conv1x1_bwd_pair_f32");
  if (rc) return rc;
  for (int j = 0; j < 2; ++j) {
    hipLaunchKernelGGL(pair_reduce_kernel, dim3((Cin[j] * 48 + 63) / 64), dim3(256), 0, st, wpartial[j], grid, Kp[j], Cin[j], dW[j]);
    rc = eml::check_launch("eml_dense_conv1x1_bwd_pair_f32(reduce)");
    if (rc) return rc;
  }
  return EML_OK;
}
