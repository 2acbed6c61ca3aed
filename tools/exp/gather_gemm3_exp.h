// Experiment record (NOT product): the wave-specialised form of gather_gemm2_kernel measured in round 4 and dropped
// (110-115 TF/s against 125-131 for the interleaved kernel on the same box, profiles/r04_gg2_variants.jsonl).
#pragma once
#include "../../emlight_amd/csrc/gather_gemm2.h"

namespace gg2 {

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form (round 4, after the measurements below): the SAME tile, tables, LDS layout and DMAs, but the
// workgroup is 8 waves -- waves 0..3 only read fragments and issue MFMAs, waves 4..7 only stage the next chunk.
// Why: measured on the 128 -> 128 layer at 128x256 (tools/exp/gg2_bench.py, same box, TF/s): the MFMA + ds_read + barrier
// loop alone 142.9; + the B-tile DMAs 129.2; + the gathered loads (no DMA) 126.6; both 118.0 -- whatever the lane mapping
// of the gather (full lines 118.0, half lines 119.0), wherever the DMAs sit in the MFMA stream (+2 %), 4 or 8 waves per
// tile (-5 %), packed or scalar combine.  Every vector-memory instruction a wave issues holds THAT wave ~60 cycles
// (MI355X_MICROARCH.md: LDS-DMA issue cost), 21 of them per 128 MFMAs; with two MFMA waves per SIMD the matrix pipe sits
// idle whenever both are held (45 cycles per instruction, measured).  Round 2's specialisation attempt lost to register
// spills in the producers (operand + table entries + the dense operand in registers at 128 VGPRs); with the dense operand
// and the table moved to LDS-DMA a producer needs ~100.
template <int BN, int LPP, bool ONE>
__global__ __launch_bounds__(512, 4) void gather_gemm3_kernel(
    const float* __restrict__ X, const int* __restrict__ idx, const float* __restrict__ wgt,
    const float* __restrict__ W2 /*[O][9C]*/, const float* __restrict__ bias, float* __restrict__ Y /*[M][O]*/, int M,
    int HW, int Po, int C, int O, int ke, const unsigned char* __restrict__ rowmax, const float* __restrict__ res,
    float slope) {
  static_assert((BN == 64 || BN == 128) && (LPP == 8 || LPP == 4), "config");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [2][kBM][kLdA]
  float* Bs = As + 2 * kBM * kLdA;                    // [2][BN][kLdB]
  float* Tab = Bs + 2 * BN * kLdB;                    // [2][ idx[kBM][4] | wgt[kBM][4] ]
  constexpr int kTab = 2 * kBM * 4;
  constexpr int WN = BN / 2, NI = WN / 16;            // consumer wave tile 64 x WN (2 x 2 waves)
  constexpr int NP = 256;                             // producer threads
  constexpr int PPT = kBM * LPP / NP, PCS = 8 / LPP, NE = ONE ? 1 : 4, NA = PPT * PCS * NE;
  constexpr int NBD = BN * kBK * 4 / 1024 / 4;        // B DMA instructions per producer wave and chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_ot = O / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt_local = slot / n_ot, ot = slot - mt_local * n_ot;
  const int mt = xcd * (int)(gridDim.x / (8 * n_ot)) + mt_local;
  if (mt * kBM >= M) return;
  const int m0 = mt * kBM, o0 = ot * BN;
  const int cpt = C / kBK;

  // virtual taps: pole-row tiles of a ke = 8 table walk every tap twice (entries 0..3, 4..7); decided by all 8 waves
  const int wq = wave & 3;
  const int tpix = min(m0 + 64 * (wq & 1) + lane, M - 1) % Po;
  bool ng2 = false;
  if (!ONE && ke == 8) {
    const bool mine = rowmax ? rowmax[tpix] > 4 : true;
    const bool any = __builtin_amdgcn_ballot_w64(mine) != 0;
    if (lane == 0) reinterpret_cast<int*>(smem)[wave] = any ? 1 : 0;
    __syncthreads();
    int f = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) f |= reinterpret_cast<const int*>(smem)[w];
    ng2 = f != 0;
    __syncthreads();
  }
  const int ngrp = __builtin_amdgcn_readfirstlane(ng2 ? 2 : 1);
  const int nvt = 9 * ngrp, nchunks = nvt * cpt;

  if (wave >= 4) {
    // ================================================================================== producers (waves 4..7)
    const int pw = wave - 4, ptid = tid - 256;
    const int gl = ptid % LPP, gp = ptid / LPP;
    const int sb0 = m0 / Po;
    const unsigned c4 = 4u * (unsigned)C;
    unsigned poff[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int m = min(m0 + gp + (NP / LPP) * u, M - 1);
      poff[u] = (unsigned)(m / Po - sb0) * (unsigned)HW * c4 + 16u * gl;
    }
    const char* xbase = reinterpret_cast<const char*>(X + (size_t)sb0 * HW * C);
    const char* tsrc = (wq & 2) ? (const char*)(wgt + (size_t)tpix * 9 * ke) : (const char*)(idx + (size_t)tpix * 9 * ke);
    auto lds_dma = [&](const void* gsrc, const float* lds_dst, auto size_tag) {
      unsigned keep;
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst);
      if constexpr (decltype(size_tag)::value == 16)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    auto table_dma = [&](int tap, int grp, int par) {
      const float* dst = Tab + par * kTab + (wq & 2) * (kBM * 2) + 64 * (wq & 1) * NE;
      if constexpr (ONE) lds_dma(tsrc + 4 * tap, dst, std::integral_constant<int, 4>{});
      else lds_dma(tsrc + 4 * (ke * tap + 4 * grp), dst, std::integral_constant<int, 16>{});
    };
    const int brow = lane >> 3, bslot = lane & 7;
    const unsigned wlane = (unsigned)(8 * NBD * pw + brow) * 9u * c4 + 16u * (unsigned)(bslot ^ bswz(brow));
    const char* wbase = reinterpret_cast<const char*>(W2 + (size_t)o0 * 9 * C);
    auto b_dma = [&](int q, int tap, int c0, int buf) {
      const float* dst = Bs + (size_t)buf * BN * kLdB + 8 * (NBD * pw + q) * kLdB;
      const char* src = wbase + ((size_t)q * 8 * 9 * C + (size_t)tap * C + c0) * 4;
      lds_dma(src + wlane, dst, std::integral_constant<int, 16>{});
    };
    // stage chunk (c, tap, grp) [table parity par] into LDS buffer buf; also sends the NEXT virtual tap's table
    auto stage = [&](int c, int tap, int grp, int par, int buf) {
      int t_tap = tap, t_grp = grp + 1;
      if (t_grp == ngrp) { t_grp = 0; t_tap = tap + 1; }
      if (t_tap > 8) { t_tap = 8; t_grp = ngrp - 1; }
      table_dma(t_tap, t_grp, par ^ 1);
#pragma unroll
      for (int q = 0; q < NBD; ++q) b_dma(q, tap, c * kBK, buf);
      const float* tb = Tab + par * kTab;
      const char* cbase = xbase + c * (kBK * 4);
      float4 av[NA];
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int px = gp + (NP / LPP) * u;
        int4 id;
        if constexpr (ONE) id = make_int4(__builtin_bit_cast(int, tb[px]), 0, 0, 0);
        else id = *reinterpret_cast<const int4*>(tb + 4 * px);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int ie = e == 0 ? id.x : e == 1 ? id.y : e == 2 ? id.z : id.w;
          const unsigned off = __umul24((unsigned)max(ie, 0), c4) + poff[u];
#pragma unroll
          for (int s2 = 0; s2 < PCS; ++s2)
            av[(u * NE + e) * PCS + s2] = *reinterpret_cast<const float4*>(cbase + off + 64 * s2);
        }
      }
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int px = gp + (NP / LPP) * u;
        float4 w;
        if constexpr (ONE) w = make_float4(tb[kBM * 4 + px], 0.f, 0.f, 0.f);
        else w = *reinterpret_cast<const float4*>(tb + kBM * 4 + 4 * px);
        float* ad = As + (size_t)buf * kBM * kLdA + px * kLdA + 4 * gl;
#pragma unroll
        for (int s2 = 0; s2 < PCS; ++s2) {
          float4 o;
          if constexpr (ONE) {
            const float4 v = av[(u * NE) * PCS + s2];
            o = make_float4(v.x * w.x, v.y * w.x, v.z * w.x, v.w * w.x);
          } else {   // grid_sample's order: nw, ne, sw, se
            const float4 v0 = av[(u * NE + 0) * PCS + s2], v1 = av[(u * NE + 1) * PCS + s2];
            const float4 v2 = av[(u * NE + 2) * PCS + s2], v3 = av[(u * NE + 3) * PCS + s2];
            o.x = fmaf(v3.x, w.w, fmaf(v2.x, w.z, fmaf(v1.x, w.y, v0.x * w.x)));
            o.y = fmaf(v3.y, w.w, fmaf(v2.y, w.z, fmaf(v1.y, w.y, v0.y * w.x)));
            o.z = fmaf(v3.z, w.w, fmaf(v2.z, w.z, fmaf(v1.z, w.y, v0.z * w.x)));
            o.w = fmaf(v3.w, w.w, fmaf(v2.w, w.z, fmaf(v1.w, w.y, v0.w * w.x)));
          }
          *reinterpret_cast<float4*>(ad + 16 * s2) = o;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMAs (B tile, next table) have landed
    };
    // prologue: the table of virtual tap 0, then chunk 0
    table_dma(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // barrier P0: table 0 visible to every producer
    stage(0, 0, 0, 0, 0);
    __syncthreads();                                    // barrier P1: chunk 0 staged
    int n_c = 1, n_tap = 0, n_grp = 0, n_par = 0;       // position of the chunk staged next (cpt >= 2)
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk + 1 < nchunks) stage(n_c, n_tap, n_grp, n_par, (chunk & 1) ^ 1);
      if (++n_c == cpt) {
        n_c = 0;
        n_par ^= 1;
        if (++n_grp == ngrp) { n_grp = 0; ++n_tap; }
      }
      eml::lds_barrier();
    }
    return;
  }
  // ==================================================================================== consumers (waves 0..3)
  const int r = lane & 15, kk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                      // P0
  __syncthreads();                                      // P1
  const int sw = bswz(r);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    const float* ab = As + (size_t)buf * kBM * kLdA + (64 * wm + r) * kLdA + 8 * kk;
    const float* bb = Bs + (size_t)buf * BN * kLdB + (WN * wn + r) * kLdB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 af[4], bf[NI];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const float4*>(ab + 16 * mi * kLdA + 4 * h);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const float4*>(bb + 16 * ni * kLdB + 4 * ((2 * kk + h) ^ sw));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(f4c(bf[ni], t), f4c(af[mi], t), acc[ni][mi]);
    }
    eml::lds_barrier();
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int o = o0 + WN * wn + 16 * ni + 4 * kk;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bq = *reinterpret_cast<const float4*>(bias + o);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + 64 * wm + 16 * mi + r;
      if (m < M) {
        float4 v = make_float4(acc[ni][mi][0] + bq.x, acc[ni][mi][1] + bq.y, acc[ni][mi][2] + bq.z, acc[ni][mi][3] + bq.w);
        if (res) {
          const float4 q = *reinterpret_cast<const float4*>(res + (size_t)m * O + o);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        *reinterpret_cast<float4*>(Y + (size_t)m * O + o) = v;
      }
    }
  }
}

}  // namespace gg2
