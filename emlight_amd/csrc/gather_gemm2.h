// Gather-GEMM, second generation (round 4): the forward / input-gradient kernel of the fused SphereConv2D
//     Y[m][o] = act(bias[o] + res[m][o] + sum_{tap,c} Ag[m][tap][c] * W2[o][tap*C + c]),
//     Ag[m][tap][c] = sum_{e < ke} wgt[p,tap,e] * X[b][idx[p,tap,e]][c]          (reference: sphere_cnn.py:111-124)
// Same 128-pixel x {64,128,256}-channel x 32 tile of v_mfma_f32_16x16x4_f32 as round 2's kernel (sphere_conv_fused.hip),
// with the three things the round-3 profile said its staging lost time on rebuilt:
//
//  1. FULL-LINE GATHERS.  Round 2 gave a pixel's 32-channel chunk (128 B = one cache line) to 2 lanes, each loading 4
//     float4 of its 64-byte half: one wave instruction touched 32 lines and 64 half-line segments for 1 KB of data, and
//     the texture-address unit -- not HBM, not L2 -- was what the waves queued on.  Here LPP = 8 consecutive lanes load the
//     8 float4 of ONE line in ONE instruction (8 lines, 16 segments per 1 KB: the minimum), and a thread owns 4 pixels
//     (2 with 512 threads) x 1 float4 instead of 1 pixel x 4 float4.  The LDS commit is then a 128-byte contiguous
//     ds_write_b128 run per pixel (conflict-free in 8-lane groups) into the same [pixel][36] tile.
//  2. THE DENSE OPERAND BY LDS-DMA.  W2's tile goes HBM/L2 -> LDS with global_load_lds_dwordx4: no VGPRs, no ds_write,
//     no VALU.  The DMA writes 1 KB lane-linear, so the tile is unpadded [o][32] with the 16-byte slots of a row XOR-
//     swizzled on the SOURCE side (slot ^ f(o), f(r) = r1 | r2 << 2: found by search, conflict-free for the fragment
//     ds_read_b128 of all four 16-lane service groups).
//  3. THE TAP TABLE IN LDS.  A thread with 4 pixels would need 64 VGPRs of table entries (current + prefetched tap); the
//     entries of a tap (128 pixels x (4 indices + 4 weights)) are DMA'd into a double-buffered 8 KB LDS table one tap
//     ahead and read back (broadcast ds_read_b128) right where they are used.
//  Pole rows of the transposed table (ke = 8, < 3 % of the tiles) run as TWO virtual taps of 4 entries over the same W2
//  chunk instead of a second, latency-exposed gather inside the commit.
//  4. (round 5) ROW-SHARED CORNERS (SH).  On a stride-1 sphere table the tap of pixel (r, c) samples source column
//     c + const(r, tap) of rows y0(r, tap), y0 + 1 (sphere_cnn.py:31-58: new_theta = theta + f(phi, tap)), so the north-east
//     corner of a pixel IS the north-west corner of its right neighbour, likewise south-east / south-west -- modulo W
//     included.  A thread that owns 4 CONSECUTIVE pixels of a row therefore needs 2 x (4 + 1) lines instead of 4 x 4: 10
//     gathered loads per chunk instead of 16 (every vector-memory instruction costs ~45 matrix-pipe cycles here).
//     Same values, same combine order: bit-identical results.  The caller vouches for the property
//     (EML_TAP_ROWSHARE: for every p % 4 != 3, idx[p][t][1] and idx[p+1][t][0] name the same source pixel unless one of them
//     is -1 -- the zero-padded wrap-around column, weight 0 --, likewise entries 3 / 2; Po % 4 == 0; checked once per
//     geometry on the host side).  Tile row rho = 32 (px % 4) + px / 4 holds pixel px, so that the LDS
//     commit of one load slot still walks consecutive rows (stride 36 floats: conflict-free as before).
//  Addresses: one wave-uniform 64-bit base (sample of the tile's first pixel + the chunk's channel offset) in SGPRs and a
//  32-bit per-lane offset -- one v_mad per load instead of 64-bit pointer arithmetic.
#pragma once
#include <type_traits>

#include "eml_common.h"

namespace gg2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

constexpr int kBM = 128;   // pixels per tile
constexpr int kBK = 32;    // channels per K-chunk (one cache line per pixel and corner)
constexpr int kLdA = 36;   // A tile row stride (floats): conflict-free ds_read_b128 fragments
constexpr int kLdB = 32;   // B tile row stride: unpadded (LDS-DMA), XOR-swizzled slots

__device__ __forceinline__ int bswz(int r) { return ((r >> 1) & 1) | (((r >> 2) & 1) << 2); }

template <int BN, int NT>
constexpr size_t lds_bytes() {
  return (size_t)(2 * kBM * kLdA + 2 * BN * kLdB + 2 * 2 * kBM * 4) * sizeof(float);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

// BN:  output channels per workgroup (64, 128: 256 threads, two workgroups per CU; 256: 512 threads, one per CU -- a
//      gathered element is reused by twice as many MFMAs)
// LPP: lanes per gathered line (8 = full 128-byte line per instruction; 4 = 64-byte halves, for the A/B record)
// PK:  bilinear combine on register pairs (v_pk_*) or scalar fmas
// ONE: single-entry tables (ke == 1: an ordinary 3x3 convolution as a gather, the VGG19 stack)
// MOD: SPADE's modulation as the epilogue (normalization.py:101-115).  The O = 2 Cn rows of W2 are the gamma and beta heads,
//      reordered by the host so that each wave's 64 rows are 32 gamma rows followed by the 32 beta rows of the SAME channels
//      (spade_row_order below): a lane then holds gamma and beta of four channels of its pixel in two accumulator tiles and
//      writes  Y[m][c] = leaky_relu(((x - mean) * istd) * (1 + gamma) + beta)  (M, Cn) -- the (M, 2 Cn) gamma | beta tensor is
//      never stored.  `mod.gamma` (M, Cn), when given, receives gamma for the backward pass.
struct SpadeEpilogue {
  const float* x;       // the normalised map's input, pixel-major (rows, Cn); at half resolution when up2
  const float* mean;    // (Cn)
  const float* istd;    // (Cn)
  float* gamma;         // (M, Cn) or NULL
  int up2, H, W;        // destination grid (Po = H * W); up2: x lives on (H/2, W/2), nearest x2 upsample folded in
};
template <int BN, int NT, int LPP, bool PK, bool ONE, bool MOD = false, bool SH = false>
__global__ __launch_bounds__(NT, (NT == 512 && BN <= 128) ? 4 : 2) void gather_gemm2_kernel(
    const float* __restrict__ X, const int* __restrict__ idx, const float* __restrict__ wgt,
    const float* __restrict__ W2 /*[O][9C]*/, const float* __restrict__ bias, float* __restrict__ Y /*[M][O]*/, int M,
    int HW /* source pixels per sample */, int Po /* destination pixels per sample */, int C, int O, int ke,
    const unsigned char* __restrict__ rowmax, const float* __restrict__ res, float slope, SpadeEpilogue mod) {
  static_assert(!MOD || (BN == 128 && NT == 256), "the SPADE epilogue pairs tiles ni / ni + 2 of a 64-row wave tile");
  static_assert((BN == 64 || BN == 128 || BN == 256) && (NT == 256 || NT == 512) && (LPP == 8 || LPP == 4), "config");
  static_assert(NT == 512 || BN != 256, "BN = 256 needs 512 threads");
  static_assert(!SH || (LPP == 8 && NT == 256 && !ONE), "row-shared corners: 4 consecutive pixels per thread, full-line loads");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [2][kBM][kLdA]
  float* Bs = As + 2 * kBM * kLdA;                    // [2][BN][kLdB]
  float* Tab = Bs + 2 * BN * kLdB;                    // [2][ idx[kBM][4] | wgt[kBM][4] ]
  constexpr int kTab = 2 * kBM * 4;                   // floats per table buffer
  constexpr int NWN = NT / 128;                       // waves along N (2 or 4); 2 along M
  constexpr int WN = BN / NWN;                        // wave tile along N: 64 (128 / 256 threads, 256 / 512) or 32 or 16
  constexpr int NI = WN / 16;
  static_assert(NI >= 1, "wave tile");
  constexpr int PPT = kBM * LPP / NT;                 // pixels per thread (4 / 2 / 2 / 1)
  constexpr int PCS = 8 / LPP;                        // float4 pieces per (pixel, corner) and thread
  constexpr int NE = ONE ? 1 : 4;                     // table entries per virtual tap
  constexpr int NA = SH ? 2 * (PPT + 1) : PPT * PCS * NE;   // gathered float4 per thread and chunk
  constexpr int NBD = BN * kBK * 4 / 1024 / (NT / 64);   // B DMA instructions per wave and chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: SGPR, usable as an asm "s" operand
  const int r = lane & 15, kk = lane >> 4;
  const int wm = wave / NWN, wn = wave % NWN;
  // XCD-aware tile order (see sphere_conv_fused.hip): XCD k walks a contiguous band of pixel tiles, O-tiles back to back
  const int n_ot = O / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt_local = slot / n_ot, ot = slot - mt_local * n_ot;
  const int mt = xcd * (int)(gridDim.x / (8 * n_ot)) + mt_local;
  if (mt * kBM >= M) return;
  const int m0 = mt * kBM, o0 = ot * BN;

  // ---- gather roles: lane gl of LPP loads float4 piece(s) gl (+ 4) of the line; pixel gp + (NT / LPP) * u
  const int gl = tid % LPP, gp = tid / LPP;
  // tile-local pixel of load slot u, and the tile row it is committed to / the pixel a tile row stands for
  auto pix_of = [&](int u) { return SH ? PPT * gp + u : gp + (NT / LPP) * u; };
  auto row_of = [&](int px) { return SH ? 32 * (px & 3) + (px >> 2) : px; };
  auto pix_of_row = [&](int rho) { return SH ? 4 * (rho & 31) + (rho >> 5) : rho; };
  const int sb0 = m0 / Po;                             // sample of the tile's first pixel (workgroup-uniform)
  unsigned poff[PPT];                                  // per pixel, BYTES: (sample - sb0) * HW * C * 4 + the lane's piece
  const unsigned c4 = 4u * (unsigned)C;
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int m = min(m0 + pix_of(u), M - 1);
    const int sb = m / Po;
    poff[u] = (unsigned)(sb - sb0) * (unsigned)HW * c4 + 16u * gl;
  }
  const char* xbase = reinterpret_cast<const char*>(X + (size_t)sb0 * HW * C);
  // ---- table DMA roles (waves 0..3): waves 0,1 the indices, waves 2,3 the weights of pixels 64 * (wave & 1) + lane
  const int wq = wave & 3;
  const int tpix = min(m0 + 64 * (wq & 1) + lane, M - 1) % Po;
  // does any pixel of the tile have more than 4 entries for some tap?  (Through the dynamic LDS array: a second
  // __shared__ object -- __syncthreads_or's -- makes hipcc drain the VM counter in front of LDS reads next to LDS-DMAs.)
  bool ng2 = false;
  if (!ONE && ke == 8) {
    const bool mine = rowmax ? rowmax[tpix] > 4 : true;
    const bool any = __builtin_amdgcn_ballot_w64(mine) != 0;
    if (lane == 0) reinterpret_cast<int*>(smem)[wave] = any ? 1 : 0;
    __syncthreads();
    int f = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) f |= reinterpret_cast<const int*>(smem)[w];
    ng2 = f != 0;
    __syncthreads();
  }
  const int ngrp = __builtin_amdgcn_readfirstlane(ng2 ? 2 : 1);
  const char* tsrc = (wq & 2) ? (const char*)(wgt + (size_t)tpix * 9 * ke) : (const char*)(idx + (size_t)tpix * 9 * ke);
  const int cpt = C / kBK;                             // chunks per (virtual) tap; >= 2 (launcher)
  const int nvt = 9 * ngrp;                            // virtual taps
  const int nchunks = nvt * cpt;
  // LDS-DMA through an asm statement (cdna_hip_programming.md 5.7): hipcc keeps no record of it, so it neither drains the
  // VM counter in front of every LDS read that might alias the DMA's target nor turns the counted waits of the gathered
  // loads into vmcnt(0).  Completion is counted by hand: the DMAs of a chunk are issued BEFORE its gathered loads, the VM
  // counter retires in order, so the wait of the last commit covers them, and an explicit vmcnt(0) closes the chunk.
  auto lds_dma = [&](const void* gsrc, const float* lds_dst, auto size_tag) {
    unsigned keep;
    // LDS byte address of the wave's 1 KB (256 B) window; uniform by construction, readfirstlane makes it provably so
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst);
    if constexpr (decltype(size_tag)::value == 16)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  // entries of virtual tap (tap, grp) -> Tab[par]; every wave takes part (waves 4..7 of the 512-thread tile repeat the
  // transfers of waves 0..3: same bytes to the same place), so the chunk loop stays one basic block
  auto table_dma = [&](int tap, int grp, int par) {
    const float* dst = Tab + par * kTab + (wq & 2) * (kBM * 2) + 64 * (wq & 1) * NE;   // wave-uniform
    if constexpr (ONE) lds_dma(tsrc + 4 * tap, dst, std::integral_constant<int, 4>{});
    else lds_dma(tsrc + 4 * (ke * tap + 4 * grp), dst, std::integral_constant<int, 16>{});
  };
  // ---- B DMA roles: instruction q of wave w covers rows 8 * (NBD * w + q) .. + 7; lane -> (row l >> 3, physical slot l & 7)
  // (the swizzle reads row bits 1, 2 only: it depends on the lane's row within the 8-row instruction, not on q)
  const int brow = lane >> 3, bslot = lane & 7;
  const unsigned wlane = (unsigned)(8 * NBD * wave + brow) * 9u * c4 + 16u * (unsigned)(bslot ^ bswz(brow));
  const char* wbase = reinterpret_cast<const char*>(W2 + (size_t)o0 * 9 * C);
  auto b_dma = [&](int q, int tap, int c0, int buf) {
    const float* dst = Bs + (size_t)buf * BN * kLdB + 8 * (NBD * wave + q) * kLdB;   // wave-uniform
    const char* src = wbase + ((size_t)q * 8 * 9 * C + (size_t)tap * C + c0) * 4;    // wave-uniform part
    lds_dma(src + wlane, dst, std::integral_constant<int, 16>{});
  };

  float4 av[NA];                                       // in flight: [pixel u][entry e][piece s]
  int4 ids[PPT];
  float4 wv[PPT];
  auto read_ids = [&](int u, int par) {
    const float* tb = Tab + par * kTab;
    const int px = pix_of(u);
    if constexpr (ONE) ids[u] = make_int4(__builtin_bit_cast(int, tb[px]), 0, 0, 0);
    else ids[u] = *reinterpret_cast<const int4*>(tb + 4 * px);
  };
  auto read_wgt = [&](int u, int par) {
    const float* tb = Tab + par * kTab + kBM * 4;
    const int px = pix_of(u);
    if constexpr (ONE) wv[u] = make_float4(tb[px], 0.f, 0.f, 0.f);
    else wv[u] = *reinterpret_cast<const float4*>(tb + 4 * px);
  };
  auto load_piece = [&](int piece, const char* cbase) {   // piece = (u * NE + e) * PCS + s
    if constexpr (SH) {
      // piece = 2 j + row: column j of the thread's PPT + 1 source columns (j < PPT: the west corners of pixel j;
      // j == PPT: the east corners of the last pixel), row 0 = north, 1 = south
      // A column shared by two pixels takes whichever of the two entries is on the map: grid_sample ZERO-PADS the column
      // that wraps around (index -1, weight 0 on that side; sphere_cnn.py:55,122), so at one seam per row and tap only one
      // of "east of j - 1" / "west of j" is a real pixel -- max() picks it, the padded side multiplies it by weight 0.
      const int j = piece >> 1, row = piece & 1, u = j < PPT ? j : PPT - 1;
      int id = j < PPT ? (row ? ids[u].z : ids[u].x) : (row ? ids[u].w : ids[u].y);
      if (j > 0 && j < PPT) id = max(id, row ? ids[j - 1].w : ids[j - 1].y);
      const unsigned off = __umul24((unsigned)max(id, 0), c4) + poff[u];
      av[piece] = *reinterpret_cast<const float4*>(cbase + off);
      return;
    }
    const int u = piece / (NE * PCS), e = (piece / PCS) % NE, s = piece % PCS;
    const int id = e == 0 ? ids[u].x : e == 1 ? ids[u].y : e == 2 ? ids[u].z : ids[u].w;
    // out-of-bounds corners (-1) carry weight 0: any valid address will do.  24-bit multiply: id < 2^24, 4C < 2^24
    const unsigned off = __umul24((unsigned)max(id, 0), c4) + poff[u];
    av[piece] = *reinterpret_cast<const float4*>(cbase + off + 64 * s);
  };
  auto commit_pixel = [&](int u, int buf) {
    const int px = pix_of(u);
    const float4 w = wv[u];
    float* ad = As + (size_t)buf * kBM * kLdA + row_of(px) * kLdA + 4 * gl;
#pragma unroll
    for (int s = 0; s < PCS; ++s) {
      float4 o;
      if constexpr (ONE) {
        const float4 v = av[(u * NE) * PCS + s];
        o = make_float4(v.x * w.x, v.y * w.x, v.z * w.x, v.w * w.x);
      } else {
        // SH: nw, ne, sw, se = columns u, u + 1 of the north / south source rows (same values as the four explicit loads)
        const float4 v0 = SH ? av[2 * u] : av[(u * NE + 0) * PCS + s], v1 = SH ? av[2 * u + 2] : av[(u * NE + 1) * PCS + s];
        const float4 v2 = SH ? av[2 * u + 1] : av[(u * NE + 2) * PCS + s], v3 = SH ? av[2 * u + 3] : av[(u * NE + 3) * PCS + s];
        if constexpr (PK) {
          typedef float v2f __attribute__((ext_vector_type(2)));
          v2f lo = v2f{v0.x, v0.y} * v2f{w.x, w.x}, hi = v2f{v0.z, v0.w} * v2f{w.x, w.x};
          lo += v2f{v1.x, v1.y} * v2f{w.y, w.y};
          hi += v2f{v1.z, v1.w} * v2f{w.y, w.y};
          lo += v2f{v2.x, v2.y} * v2f{w.z, w.z};
          hi += v2f{v2.z, v2.w} * v2f{w.z, w.z};
          lo += v2f{v3.x, v3.y} * v2f{w.w, w.w};
          hi += v2f{v3.z, v3.w} * v2f{w.w, w.w};
          o = make_float4(lo.x, lo.y, hi.x, hi.y);
        } else {   // grid_sample's order: nw, ne, sw, se
          o.x = fmaf(v3.x, w.w, fmaf(v2.x, w.z, fmaf(v1.x, w.y, v0.x * w.x)));
          o.y = fmaf(v3.y, w.w, fmaf(v2.y, w.z, fmaf(v1.y, w.y, v0.y * w.x)));
          o.z = fmaf(v3.z, w.w, fmaf(v2.z, w.z, fmaf(v1.z, w.y, v0.z * w.x)));
          o.w = fmaf(v3.w, w.w, fmaf(v2.w, w.z, fmaf(v1.w, w.y, v0.w * w.x)));
        }
      }
      *reinterpret_cast<float4*>(ad + 16 * s) = o;
    }
  };

  f32x4 acc[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: table of virtual tap 0, then chunk 0 staged synchronously
  table_dma(0, 0, 0);
#pragma unroll
  for (int q = 0; q < NBD; ++q) b_dma(q, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PPT; ++u) read_ids(u, 0);
#pragma unroll
  for (int piece = 0; piece < NA; ++piece) load_piece(piece, xbase);
#pragma unroll
  for (int u = 0; u < PPT; ++u) read_wgt(u, 0);
#pragma unroll
  for (int u = 0; u < PPT; ++u) commit_pixel(u, 0);
  __syncthreads();

  constexpr int kHalf = 4 * NI * 4;                    // MFMAs per K-half and wave: 64 (WN = 64) or 32
  constexpr int kVm = NA;                              // gathered loads per chunk (the DMAs go first, in asm statements)
  static_assert(kVm <= kHalf, "more gathered loads than MFMAs in a K-half");
  constexpr int kEvery = kHalf / kVm;                  // MFMAs per gathered load in half 0
  constexpr int kC0 = kHalf / 4, kCE = (kHalf - kC0) / PPT;   // half 1: first commit behind MFMA kC0, then every kCE
  constexpr int kDE = kC0 / (1 + NBD) > 0 ? kC0 / (1 + NBD) : 1;   // half 1: a DMA behind every kDE-th MFMA, before the commits

  // position of the NEXT chunk (the one being staged), kept incrementally -- no divisions in the loop:
  // (cpt >= 2, i.e. C >= 64, is a launcher precondition: chunk 1 then still belongs to virtual tap 0)
  int n_c = 1, n_tap = 0, n_grp = 0, n_par = 0;        // chunk within the virtual tap, tap, group, table parity
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    const bool last = chunk + 1 == nchunks;
    // the last chunk re-stages its own operands into the idle buffer (loads stay unconditional)
    const int s_c = last ? cpt - 1 : n_c, s_tap = last ? 8 : n_tap, s_grp = last ? ngrp - 1 : n_grp;
    const int s_par = last ? ((nvt - 1) & 1) : n_par;
    const char* cbase = xbase + s_c * (kBK * 4);
    // the table of the virtual tap AFTER the staged one (clamped to the last), into the OTHER parity -- the buffer nobody
    // reads during this chunk: its last readers staged the previous virtual tap, a barrier ago.  Re-sent by every chunk of
    // a tap (same bytes: the loop body stays free of branches), first read a barrier after its first transfer.
    int t_tap = s_tap, t_grp = s_grp + 1;
    if (t_grp == ngrp) { t_grp = 0; t_tap = s_tap + 1; }
    if (t_tap > 8) { t_tap = 8; t_grp = ngrp - 1; }
    __builtin_amdgcn_sched_barrier(0);
    const float* ab = As + (size_t)buf * kBM * kLdA + (64 * wm + r) * kLdA + 8 * kk;
    const float* bb = Bs + (size_t)buf * BN * kLdB + (WN * wn + r) * kLdB;
    const int sw = bswz(r);
    // DMAs of the staged chunk first (asm statements: they keep their place in front of the loads below)
#if !defined(GG2_NODMA) && defined(GG2_DMA_EARLY)
    table_dma(t_tap, t_grp, s_par ^ 1);
#pragma unroll
    for (int q = 0; q < NBD; ++q) b_dma(q, s_tap, s_c * kBK, buf ^ 1);
#endif
#ifndef GG2_NOTAB
#pragma unroll
    for (int u = 0; u < PPT; ++u) read_ids(u, s_par);
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 af[4], bf[NI];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const float4*>(ab + 16 * mi * kLdA + 4 * h);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const float4*>(bb + 16 * ni * kLdB + 4 * ((2 * kk + h) ^ sw));
#ifndef GG2_NOTAB
      if (h == 1) {   // the bilinear weights of the pixels committed in this half
#pragma unroll
        for (int u = 0; u < PPT; ++u) read_wgt(u, s_par);
      }
#endif
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            acc[ni][mi] = mfma16(f4c(bf[ni], t), f4c(af[mi], t), acc[ni][mi]);
            const int cnt = (t * NI + ni) * 4 + mi;    // MFMA index within this K-half
#if !defined(GG2_NODMA) && !defined(GG2_DMA_EARLY)
            // the DMAs of the staged chunk (table, then the B tile), one behind every kDE-th MFMA at the start of half 1:
            // issued back to back at the top of the chunk they held the wave ~100 cycles each with the matrix pipe waiting
            // (measured: 141 -> 126 TF/s for the loop without any A staging).  They are YOUNGER than the gathered loads,
            // so the compiler's counted waits of the commits only get stricter; the vmcnt(0) that closes the chunk covers them.
            if (h == 1 && cnt % kDE == 0 && cnt / kDE < 1 + NBD) {
              if (cnt == 0) table_dma(t_tap, t_grp, s_par ^ 1);
              else b_dma(cnt / kDE - 1, s_tap, s_c * kBK, buf ^ 1);
              __builtin_amdgcn_sched_barrier(0);
            }
#endif
#ifdef GG2_NOGATHER   // experiment build: no A staging at all (wrong results; the ceiling of the B-DMA + MFMA loop)
            continue;
#endif
            if (h == 0) {
              if (cnt % kEvery == kEvery - 1 && cnt / kEvery < kVm) load_piece(cnt / kEvery, cbase);
            } else if (cnt >= kC0 && (cnt - kC0) % kCE == 0 && (cnt - kC0) / kCE < PPT) {
              // half 1: each commit pinned where it stands (the DAG order of this half is kept by the scheduler)
              commit_pixel((cnt - kC0) / kCE, buf ^ 1);
#ifdef GG2_SETPRIO
              if ((cnt - kC0) / kCE == PPT - 1) __builtin_amdgcn_s_setprio(2);
#endif
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      if (h == 0) {
        // ---- half 0 as a pipeline for the machine scheduler (one scheduling region; the pre-RA DAG order is NOT kept for
        // the loads -- under register pressure it sinks all of them behind the MFMAs).  Masks: 0x8 MFMA, 0x20 VMEM read,
        // 0x100 DS read: fragments + table ids, then one gathered load behind every kEvery MFMAs
#ifndef GG2_NOGATHER
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + NI + PPT, 0);
#pragma unroll
        for (int i = 0; i < kVm; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, kEvery, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        if (kHalf - kVm * kEvery > 0) __builtin_amdgcn_sched_group_barrier(0x008, kHalf - kVm * kEvery, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // advance the staged position
    if (++n_c == cpt) {
      n_c = 0;
      n_par ^= 1;
      if (++n_grp == ngrp) { n_grp = 0; ++n_tap; }
    }
    // the DMAs of this chunk (B tile, table) have landed once the wave's VM counter is drained: every gathered load
    // issued after them has been consumed by the commits above
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GG2_SETPRIO    // experiment build: the tail of a chunk (MFMAs only) runs at raised priority
    __builtin_amdgcn_s_setprio(0);
#endif
#ifndef GG2_NOBAR
    eml::lds_barrier();
#endif
  }
  if constexpr (MOD) {
    // ---- SPADE epilogue: tiles ni = 0, 1 hold gamma, ni + 2 beta of channels cb + 16 ni + 4kk .. +3; the arithmetic is
    // spade_norm_modulate_fwd_kernel's (csrc/spade.hip), expression for expression
    const int Cn = O >> 1;
    const int cb = (o0 >> 1) + 32 * wn;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int c = cb + 16 * ni + 4 * kk;
      const int og = o0 + 64 * wn + 16 * ni + 4 * kk;      // row of the (reordered) bias: gamma, + 32: beta
      float4 bg = make_float4(0.f, 0.f, 0.f, 0.f), bb = bg;
      if (bias) {
        bg = *reinterpret_cast<const float4*>(bias + og);
        bb = *reinterpret_cast<const float4*>(bias + og + 32);
      }
      const float4 mu = *reinterpret_cast<const float4*>(mod.mean + c);
      const float4 is = *reinterpret_cast<const float4*>(mod.istd + c);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + pix_of_row(64 * wm + 16 * mi + r);
        if (m < M) {
          size_t xrow = (size_t)m;
          if (mod.up2) {
            const unsigned b = (unsigned)m / (unsigned)Po, p = (unsigned)m - b * (unsigned)Po;
            const unsigned h = p / (unsigned)mod.W, w = p - h * (unsigned)mod.W;
            xrow = ((size_t)b * (unsigned)(mod.H >> 1) + (h >> 1)) * (unsigned)(mod.W >> 1) + (w >> 1);
          }
          const float4 a = *reinterpret_cast<const float4*>(mod.x + xrow * Cn + c);
          const float4 g = make_float4(acc[ni][mi][0] + bg.x, acc[ni][mi][1] + bg.y, acc[ni][mi][2] + bg.z, acc[ni][mi][3] + bg.w);
          const float4 be = make_float4(acc[ni + 2][mi][0] + bb.x, acc[ni + 2][mi][1] + bb.y, acc[ni + 2][mi][2] + bb.z,
                                        acc[ni + 2][mi][3] + bb.w);
          float4 v;
          v.x = fmaf((a.x - mu.x) * is.x, 1.f + g.x, be.x);
          v.y = fmaf((a.y - mu.y) * is.y, 1.f + g.y, be.y);
          v.z = fmaf((a.z - mu.z) * is.z, 1.f + g.z, be.z);
          v.w = fmaf((a.w - mu.w) * is.w, 1.f + g.w, be.w);
          v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
          v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
          *reinterpret_cast<float4*>(Y + (size_t)m * Cn + c) = v;
          if (mod.gamma) *reinterpret_cast<float4*>(mod.gamma + (size_t)m * Cn + c) = g;
        }
      }
    }
    return;
  }
  // ---- epilogue: lane (r, kk) owns output channels 4kk..4kk+3 of tile ni for pixel r of tile mi
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int o = o0 + WN * wn + 16 * ni + 4 * kk;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bq = *reinterpret_cast<const float4*>(bias + o);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + pix_of_row(64 * wm + 16 * mi + r);
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
