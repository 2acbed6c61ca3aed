// Gather-GEMM, third generation (round 6): the SphereConv2D product of the LOW-RESOLUTION, WIDE layers
//     Y[m][o] = act(bias[o] + res[m][o] + sum_{tap,c} Ag[m][tap][c] * W2[o][tap*C + c]),
//     Ag[m][tap][c] = sum_{e < ke} wgt[p,tap,e] * X[b][idx[p,tap,e]][c]          (reference: sphere_cnn.py:111-124)
// for the layers rounds 2-5 left on sphere_im2col + a library GEMM (+ sphere_col2im on the way back): 1024 -> 1024 @8x16,
// 128 -> 2048 / 1024 -> 512 @16x32, 128 -> 1024 / 512 -> 256 @32x64 ... -- few pixels, many channels, K = 9 C long.
//
// What is different from gather_gemm2.h (the kernel of the high-resolution layers):
//
//  1. THE SOURCE FOOTPRINT LIVES IN LDS.  On these grids the 9 taps of a 128-pixel tile touch a handful of source ROWS
//     (8 x 16: the whole sample, 128 pixels; 16 x 32: <= 8 rows = 256 pixels; 32 x 64: <= 7 rows = 448): a contiguous
//     pixel range [lo, lo + n) of the sample, found per tile on the host from the tap table itself.  K runs (channel chunk
//     outer, tap inner): a 32-channel chunk of the footprint (n x 128 B) is staged ONCE -- plain, contiguous, full-line
//     global loads into registers a chunk ahead, one ds_write burst at the chunk boundary -- and serves all 9 taps.  No
//     gathered global load exists: gg2 fetches 4 corners x 9 taps = 36 lines per pixel and chunk through the texture path
//     (its misses and instruction count were 11 % of its rate: DESIGN 10.3), this kernel n / 128 <= 3.5.
//  2. NO A TILE.  A lane builds its MFMA fragment directly: 4 ds_read_b128 (the four bilinear corners of ITS pixel, 4 of
//     its 8 channels of the K-chunk) and 16 FMAs per fragment, in grid_sample's order -- the same arithmetic as gg2's
//     commit, so the two kernels agree bit for bit.  No ds_write of an operand tile, no second pass over it.
//  3. PER-WAVE TAP TABLE.  A wave's 32 pixels need 32 x (4 indices + 4 weights) = 1 KB per tap: ONE LDS-DMA instruction
//     into a wave-private double buffer (lanes 0-31 the indices, 32-63 the weights).  Private means no barrier: the
//     wave's own counted vmcnt covers it, so the fragments of the NEXT tap's first K-half are read and combined under the
//     MFMAs of this tap's second half, before the barrier that hands over the dense operand.
//  4. SPLIT-K OVER CHANNEL CHUNKS for the layers with too few pixel tiles to fill 256 CUs (1024 -> 1024 @8x16 at 32 per
//     GPU: 32 x 8 tiles): raw accumulators to [split][M][O], then gg3_reduce_kernel sums them in a fixed order and applies
//     the epilogue -- deterministic, no atomics.
// Unchanged from gg2: the dense operand W2 by LDS-DMA into an unpadded, XOR-swizzled [o][32] tile, double-buffered per
// (chunk, tap); v_mfma_f32_16x16x4_f32; pole rows of a transposed table (ke = 8) as two virtual taps; the epilogue.
// Wave tile: 32 pixels x 128 output channels (2 x 8 accumulator tiles); 256 threads = 128 x 128, two workgroups per CU
// (76 KB of LDS each); 512 threads = 128 x 256, one per CU (143 KB: the 448-pixel footprint of the 32 x 64 grids).
#pragma once
#include <type_traits>

#include "eml_common.h"

namespace gg3 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kBM = 128;   // pixels per tile
constexpr int kBK = 32;    // channels per K-chunk
constexpr int kLdF = 36;   // footprint row stride (floats): 144 B -- the corner reads of consecutive pixels spread over the banks
constexpr int kLdB = 32;   // B tile row stride: unpadded (LDS-DMA), XOR-swizzled slots

__device__ __forceinline__ int bswz(int r) { return ((r >> 1) & 1) | (((r >> 2) & 1) << 2); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

template <int BN, int NT, int FPX>
constexpr size_t lds_bytes() {
  return (size_t)(FPX * kLdF + 2 * BN * kLdB + (NT / 64) * 512) * sizeof(float);
}

// BN: output channels per workgroup (128 with 256 threads, 256 with 512); FPX: footprint capacity in source pixels
template <int BN, int NT, int FPX>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void gather_gemm3_kernel(
    const float* __restrict__ X, const int* __restrict__ idx, const float* __restrict__ wgt,
    const float* __restrict__ W2 /*[O][9C]*/, const float* __restrict__ bias, float* __restrict__ Y /*[M][O] or [split][M][O]*/,
    int M, int HW /* source pixels per sample */, int Po /* destination pixels per sample */, int C, int O, int ke,
    const unsigned char* __restrict__ rowmax, const float* __restrict__ res, float slope,
    const int* __restrict__ fp /* (lo, n) per 128-pixel tile of a sample; NULL: Po < 128, whole samples */, int nsplit) {
  static_assert((BN == 128 && NT == 256) || (BN == 256 && NT == 512), "config");
  static_assert(FPX % 32 == 0 && (FPX * 8) % NT == 0, "footprint pieces per thread");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Fp = smem;                                   // [FPX][kLdF]
  float* Bs = Fp + FPX * kLdF;                        // [2][BN][kLdB]
  float* Tab = Bs + 2 * BN * kLdB;                    // [waves][2][ idx[32][4] | wgt[32][4] ]
  constexpr int NW = NT / 64, NWN = BN / 128;
  constexpr int NI = 8, MI = 2;                       // wave tile: 32 pixels x 128 channels
  constexpr int NBD = BN * kBK * 4 / 1024 / NW;       // B DMA instructions per wave and (chunk, tap): 4
  constexpr int NP = FPX * 8 / NT;                    // footprint float4 pieces per thread and chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kk = lane >> 4;
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- tile: XCD k walks a contiguous range of the (split, O-tile, pixel tile) sequence, pixel tiles fastest -- the
  // workgroups resident on an XCD at a time share their W2 slice (read once from HBM / MALL, then from that XCD's L2)
  const int n_mt = (M + kBM - 1) / kBM, n_ot = O / BN;
  const long T = (long)n_mt * n_ot * nsplit;
  const long per_xcd = (T + 7) / 8;
  const long L = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (L >= T) return;
  const int pair = (int)(L / n_mt), mt = (int)(L - (long)pair * n_mt);
  const int ot = pair % n_ot, sp = pair / n_ot;
  const int m0 = mt * kBM, o0 = ot * BN;
  const int nch = C / kBK, cps = nch / nsplit;        // launcher: nch % nsplit == 0
  const int c_begin = sp * cps, c_end = c_begin + cps;

  // ---- footprint of this tile: source pixels [lo, lo + nfp) counted from sample sb0
  const int sb0 = m0 / Po;
  int lo = 0, nfp;
  if (fp) {
    const int t = (m0 - sb0 * Po) >> 7;
    lo = fp[2 * t];
    nfp = fp[2 * t + 1];
  } else {
    const int ns = min(kBM / Po, (M - m0 + Po - 1) / Po);
    nfp = ns * HW;
  }
  const char* xfp = reinterpret_cast<const char*>(X + ((size_t)sb0 * HW + lo) * C);
  unsigned gofs[NP];                                   // BYTES from xfp; piece p = tid + j NT: pixel p >> 3, float4 p & 7
  const unsigned c4 = 4u * (unsigned)C;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int p = tid + j * NT;
    gofs[j] = (unsigned)min(p >> 3, nfp - 1) * c4 + 16u * (unsigned)(p & 7);
  }
  f32x4 pf[NP];   // (a native vector type: an array of HIP's float4 STRUCTS that lives across the tap loop stays in scratch memory)
  auto load_fp = [&](int ch) {
    const char* base = xfp + (size_t)ch * (kBK * 4);
#pragma unroll
    for (int j = 0; j < NP; ++j) pf[j] = *reinterpret_cast<const f32x4*>(base + gofs[j]);
  };
  auto store_fp = [&]() {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = tid + j * NT;
      *reinterpret_cast<f32x4*>(Fp + (p >> 3) * kLdF + 4 * (p & 7)) = pf[j];
    }
  };

  // ---- this lane's two pixels (tiles mi = 0, 1 of the wave's 32) and where their table entries point inside the footprint
  // (the table this kernel takes is FOOTPRINT-LOCAL and pre-multiplied: entry = kLdF * (source pixel - the tile's first
  // footprint pixel), 0 for an empty slot -- eml_sphere_conv_lowres_table_i32 builds it once per geometry; what is left to add
  // per tap is the sample's offset when a tile spans several samples, and the lane's 8 kk: one add per corner)
  int loc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = min(m0 + 32 * wm + 16 * mi + r, M - 1);
    loc[mi] = (fp ? 0 : (m / Po - sb0) * HW * kLdF) + 8 * kk;
  }
  // ---- table DMA role: lanes 0..31 the indices, 32..63 the weights of pixel 32 wm + (lane & 31)
  const int tpix = min(m0 + 32 * wm + (lane & 31), M - 1) % Po;
  bool ng2 = false;
  if (ke == 8) {   // does any pixel of the TILE have more than 4 entries for some tap?  (uniform over the workgroup)
    const bool mine = rowmax ? rowmax[tpix] > 4 : true;
    const bool any = __builtin_amdgcn_ballot_w64(mine) != 0;
    if (lane == 0) reinterpret_cast<int*>(smem)[wave] = any ? 1 : 0;
    __syncthreads();
    int f = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) f |= reinterpret_cast<const int*>(smem)[w];
    ng2 = f != 0;
    __syncthreads();
  }
  const int ngrp = __builtin_amdgcn_readfirstlane(ng2 ? 2 : 1);
  const int nvt = 9 * ngrp;
  const char* tsrc = (lane & 32) ? (const char*)(wgt + (size_t)tpix * 9 * ke) : (const char*)(idx + (size_t)tpix * 9 * ke);
  float* Tw = Tab + wave * 512;                        // wave-private: [2][ idx[32][4] | wgt[32][4] ]

  auto lds_dma16 = [&](const void* gsrc, const float* lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  auto table_dma = [&](int vt, int par) {
    const int tap = ngrp == 2 ? (vt >> 1) : vt, grp = ngrp == 2 ? (vt & 1) : 0;
    lds_dma16(tsrc + 4 * (ke * tap + 4 * grp), Tw + par * 256);
  };
  // ---- B DMA roles: instruction q of wave w covers rows 8 (NBD w + q) .. + 7; lane -> (row l >> 3, physical slot l & 7)
  const int brow = lane >> 3, bslot = lane & 7;
  const unsigned wlane = (unsigned)(8 * NBD * wave + brow) * 9u * c4 + 16u * (unsigned)(bslot ^ bswz(brow));
  const char* wbase = reinterpret_cast<const char*>(W2 + (size_t)o0 * 9 * C);
  auto b_dma = [&](int q, int tap, int c0, int buf) {
    const float* dst = Bs + (size_t)buf * BN * kLdB + 8 * (NBD * wave + q) * kLdB;
    const char* src = wbase + ((size_t)q * 8 * 9 * C + (size_t)tap * C + c0) * 4;
    lds_dma16(src + wlane, dst);
  };

  // ---- fragment construction
  int co[MI][4];                                       // float offsets into Fp of the four corners (+ the lane's 8 kk)
  float4 wv[MI];
  int4 tid4[MI];                                       // a tap's table entries as read, before they become offsets
  auto load_table = [&](int par) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      tid4[mi] = *reinterpret_cast<const int4*>(Tw + par * 256 + (16 * mi + r) * 4);
      wv[mi] = *reinterpret_cast<const float4*>(Tw + par * 256 + 128 + (16 * mi + r) * 4);
    }
  };
  auto table_offsets = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int4 id = tid4[mi];
      // (round 6, first build: raw indices here -- eight selects for the empty slots, which the compiler turned into divergent
      // branches inside the tap loop, and eight quarter-rate v_mul_lo_u32 by the row stride)
      co[mi][0] = id.x + loc[mi];
      co[mi][1] = id.y + loc[mi];
      co[mi][2] = id.z + loc[mi];
      co[mi][3] = id.w + loc[mi];
    }
  };
  auto read_corners = [&](float4 (&cv)[MI][4], int h) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 4; ++e) cv[mi][e] = *reinterpret_cast<const float4*>(Fp + co[mi][e] + 4 * h);
  };
  auto combine1 = [&](float4 (&af)[MI], const float4 (&cv)[MI][4], int mi) {   // one pixel tile (experiment builds place the two apart)
    const float4 w = wv[mi];
#ifndef GG3_SCALARCOMBINE   // on register pairs (v_pk_mul_f32 / v_pk_fma_f32): half the VALU instructions, same sums in the same order
                           // (experiment build GG3_SCALARCOMBINE: scalar fmaf chains; pairs + spread placement measured +2 %)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f lo = v2f{cv[mi][0].x, cv[mi][0].y} * v2f{w.x, w.x}, hi = v2f{cv[mi][0].z, cv[mi][0].w} * v2f{w.x, w.x};
    lo += v2f{cv[mi][1].x, cv[mi][1].y} * v2f{w.y, w.y};
    hi += v2f{cv[mi][1].z, cv[mi][1].w} * v2f{w.y, w.y};
    lo += v2f{cv[mi][2].x, cv[mi][2].y} * v2f{w.z, w.z};
    hi += v2f{cv[mi][2].z, cv[mi][2].w} * v2f{w.z, w.z};
    lo += v2f{cv[mi][3].x, cv[mi][3].y} * v2f{w.w, w.w};
    hi += v2f{cv[mi][3].z, cv[mi][3].w} * v2f{w.w, w.w};
    af[mi] = make_float4(lo.x, lo.y, hi.x, hi.y);
#else
    af[mi].x = fmaf(cv[mi][3].x, w.w, fmaf(cv[mi][2].x, w.z, fmaf(cv[mi][1].x, w.y, cv[mi][0].x * w.x)));
    af[mi].y = fmaf(cv[mi][3].y, w.w, fmaf(cv[mi][2].y, w.z, fmaf(cv[mi][1].y, w.y, cv[mi][0].y * w.x)));
    af[mi].z = fmaf(cv[mi][3].z, w.w, fmaf(cv[mi][2].z, w.z, fmaf(cv[mi][1].z, w.y, cv[mi][0].z * w.x)));
    af[mi].w = fmaf(cv[mi][3].w, w.w, fmaf(cv[mi][2].w, w.z, fmaf(cv[mi][1].w, w.y, cv[mi][0].w * w.x)));
#endif
  };
  auto combine = [&](float4 (&af)[MI], const float4 (&cv)[MI][4]) {   // grid_sample's order: nw, ne, sw, se (= gg2's commit)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const float4 w = wv[mi];
      af[mi].x = fmaf(cv[mi][3].x, w.w, fmaf(cv[mi][2].x, w.z, fmaf(cv[mi][1].x, w.y, cv[mi][0].x * w.x)));
      af[mi].y = fmaf(cv[mi][3].y, w.w, fmaf(cv[mi][2].y, w.z, fmaf(cv[mi][1].y, w.y, cv[mi][0].y * w.x)));
      af[mi].z = fmaf(cv[mi][3].z, w.w, fmaf(cv[mi][2].z, w.z, fmaf(cv[mi][1].z, w.y, cv[mi][0].z * w.x)));
      af[mi].w = fmaf(cv[mi][3].w, w.w, fmaf(cv[mi][2].w, w.z, fmaf(cv[mi][1].w, w.y, cv[mi][0].w * w.x)));
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: table and dense operand of (c_begin, virtual tap 0), footprint chunk c_begin into registers
  table_dma(0, 0);
#pragma unroll
  for (int q = 0; q < NBD; ++q) b_dma(q, 0, c_begin * kBK, 0);
  load_fp(c_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int g = 0;                                           // (chunk, tap) counter: parity of the B / table buffers
  float4 af0[MI], af1[MI];
  for (int ch = c_begin; ch < c_end; ++ch) {
    // every wave has passed the barrier that closed the previous chunk's last tap: nobody reads Fp any more
#ifndef GG3_NOFP
    store_fp();
    eml::lds_barrier();
    load_fp(min(ch + 1, c_end - 1));                   // the next chunk's footprint, a whole chunk (9+ taps) ahead
#endif
    {
      float4 cv[MI][4];
      load_table(g & 1);
      table_offsets();
      read_corners(cv, 0);
      combine(af0, cv);
    }
    for (int vt = 0; vt < nvt; ++vt, ++g) {
      const int buf = g & 1;
      // what is staged during this (chunk, tap): the next one; the very last re-stages itself (loads stay unconditional)
      int n_vt = vt + 1, n_ch = ch;
      if (n_vt == nvt) { n_vt = 0; n_ch = ch + 1; }
      if (n_ch == c_end) { n_vt = nvt - 1; n_ch = ch; }
      const int n_tap = ngrp == 2 ? (n_vt >> 1) : n_vt;
      __builtin_amdgcn_sched_barrier(0);
#ifndef GG3_NOTAB   // (experiment builds, tools/exp/gg3_variants.sh: pieces of the loop removed one at a time -- wrong results)
      table_dma(n_vt, buf ^ 1);                        // first: the counted wait below (vmcnt(NBD)) then covers it
#endif
      const float* bb = Bs + (size_t)buf * BN * kLdB + (128 * wn + r) * kLdB;
      const int sw = bswz(r);
      float4 cv[MI][4];
      // A K-half = 8 groups of 8 MFMAs (group = one 16-channel output tile ni: 4 k-steps x 2 pixel tiles, the two accumulators
      // alternate so that a dependent MFMA is two issues = 64 cycles behind its producer).  Everything else of the (chunk, tap)
      // is pinned behind a named group -- each piece early enough for its consumer, none bunched in front of the MFMAs:
      //   half 0: the corner reads of this tap's second K-half in front; one dense-operand DMA behind groups 0..3;
      //           combine (second half's fragments) behind group 5
      //   half 1: behind group 1 the counted wait for the next tap's table (issued a K-half ago; the NBD dense-operand DMAs
      //           behind it may still fly) and its entries -- wave-private, no barrier needed; behind group 2 the offsets;
      //           behind group 3 the next tap's first-half corners (the footprint is static within a chunk); combine behind 6
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 bf[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bf[ni] = *reinterpret_cast<const float4*>(bb + 16 * ni * kLdB + 4 * ((2 * kk + h) ^ sw));
#ifndef GG3_NOCORNER
        if (h == 0) read_corners(cv, 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[ni][mi] = mfma16(f4c(bf[ni], t), f4c(h == 0 ? af0[mi] : af1[mi], t), acc[ni][mi]);
          if (h == 0) {
#ifndef GG3_NOBDMA
            if (ni < NBD) b_dma(ni, n_tap, n_ch * kBK, buf ^ 1);
#endif
#ifndef GG3_NOCORNER
#ifndef GG3_BUNCHED   // the two pixel tiles' combines behind different MFMA groups (experiment build GG3_BUNCHED: both behind one)
            if (ni == 4) combine1(af1, cv, 0);
            if (ni == 6) combine1(af1, cv, 1);
#else
            if (ni == 5) combine(af1, cv);
#endif
#endif
          } else {
#ifndef GG3_NOTAB
            if (ni == 1) {
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBD) : "memory");
              load_table(buf ^ 1);
            }
            if (ni == 2) table_offsets();
#endif
#ifndef GG3_NOCORNER
            if (ni == 3) read_corners(cv, 0);
#ifndef GG3_BUNCHED
            if (ni == 5) combine1(af0, cv, 0);
            if (ni == 7) combine1(af0, cv, 1);
#else
            if (ni == 6) combine(af0, cv);
#endif
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef GG3_NOBAR
      eml::lds_barrier();
#endif
    }
  }

  // ---- epilogue: lane (r, kk) owns output channels 4kk..4kk+3 of tile ni for pixel r of tile mi
  if (nsplit > 1) {
    float* P = Y + (size_t)sp * M * O;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int o = o0 + 128 * wn + 16 * ni + 4 * kk;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + 32 * wm + 16 * mi + r;
        if (m < M)
          *reinterpret_cast<float4*>(P + (size_t)m * O + o) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
      }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int o = o0 + 128 * wn + 16 * ni + 4 * kk;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bq = *reinterpret_cast<const float4*>(bias + o);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + 32 * wm + 16 * mi + r;
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

// lidx[i] = kLdF * (idx[i] - lo(tile of i's destination pixel)), 0 for an empty slot (-1): the footprint-local table of the kernel
__global__ __launch_bounds__(256) void gg3_table_kernel(const int* __restrict__ idx, const int* __restrict__ fp, int* __restrict__ lidx,
                                                        long n, int per_pixel) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = idx[i];
  const int lo = fp ? fp[2 * (int)((i / per_pixel) >> 7)] : 0;
  lidx[i] = v < 0 ? 0 : (v - lo) * kLdF;
}

// Y = act(bias + res + sum_s partial[s]) in the fixed order s = 0, 1, ...: the second pass of the split-K launches
__global__ __launch_bounds__(256) void gg3_reduce_kernel(const float* __restrict__ partial, int nsplit, size_t MO, int O,
                                                         const float* __restrict__ bias, const float* __restrict__ res,
                                                         float slope, float* __restrict__ Y) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= MO) return;
  float4 v = *reinterpret_cast<const float4*>(partial + e);
  for (int s = 1; s < nsplit; ++s) {
    const float4 q = *reinterpret_cast<const float4*>(partial + (size_t)s * MO + e);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias + (e % O));
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (res) {
    const float4 q = *reinterpret_cast<const float4*>(res + e);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
  v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
  *reinterpret_cast<float4*>(Y + e) = v;
}

}  // namespace gg3
