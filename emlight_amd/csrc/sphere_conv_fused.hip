// Fused SphereConv2D for gfx950: the 9 bilinear taps of every output pixel are gathered straight into the LDS tile
// that feeds the f32 MFMAs -- the (B*H'*W', 9C) operand A9 of the unfused path (sphere_conv.hip: im2col -> library
// GEMM) is never written to HBM, neither in the forward nor for the weight gradient.
// Reference: models/networks/spherenet/sphere_cnn.py:111-124,  y = conv2d(grid_sample(x, grid), w, b, stride=3).
//
//   forward   Y[m][o]        = bias[o] + sum_{tap,c} Ag[m][tap][c] * W2[o][tap*C + c]
//   dgrad     dX[q][c]       = sum_{tap,o} Dg[q][tap][o] * W2[o][tap*C + c]     (Dg: dY gathered through the transposed
//                              table -- the forward kernel with the roles of the two pixel grids swapped)
//   wgrad     dW2[o][tap*C+c] = sum_m dY[m][o] * Ag[m][tap][c]
//   with      Ag[m][tap][c]  = sum_{k<4} wgt[p,tap,k] * X[b][idx[p,tap,k]][c],   m = b*Po + p   (the tap table of
//                              eml_sphere_tap_table_f32: grid_sample's own corners and weights; -1 = zero padding)
//
// Both are 128 x {64,128} x 32 tiled GEMMs, 256 threads = 2 x 2 waves, v_mfma_f32_16x16x4_f32 (exact f32), double-
// buffered LDS with ONE LDS-only barrier per K-chunk.  The operands of chunk i+1 (16 gathered float4 per thread + the
// dense operand) are requested ONE AT A TIME BETWEEN the MFMAs of chunk i's first K-half and committed (bilinear
// combine + ds_write) between the MFMAs of its second half; the tap-table entries are requested a whole tap ahead.
// (Requested back to back in front of the MFMA block, the loads held each wave at the texture-address unit for
// ~2000 cycles per chunk with the matrix pipe idle: 91 TF/s; interleaved: 112-115 TF/s; the bare MFMA + ds_read loop: 139.)
// Loads are unconditional from clamped addresses (zero weights / masked stores handle the borders).
//   forward: D^T form (weights = MFMA A operand, pixels = B operand): a lane owns 4 consecutive output channels of a
//            pixel -> 16-byte stores; MFMA's k index is only a summation label, so a lane's 8 consecutive floats of
//            the 32-wide K-chunk serve k-steps 0..7 (two ds_read_b128 per fragment, LDS row stride 36: conflict-free).
//   wgrad:   K = pixels; tiles keep their natural [pixel][channel] layout (row stride 144 puts the four k-rows of a
//            ds_read_b32 on disjoint bank quarters); split-K over blockIdx.z, partials + a deterministic reduction.
#include "eml_common.h"
#include "gather_gemm2.h"

#include <cstdlib>
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

constexpr int kBM = 128;   // pixels per tile (forward) / output channels per tile (wgrad)
constexpr int kBK = 32;    // K-chunk: 32 channels of one tap (forward) / 32 pixels (wgrad)
constexpr int kLdF = 36;   // forward LDS row stride (floats)
constexpr int kLdW = 144;  // wgrad LDS row stride (floats)

struct Tap {
  int4 id;
  float4 w;
};

// bilinear combination in grid_sample's order: nw, ne, sw, se.  Written on explicit register PAIRS (x,y) / (z,w) of
// each loaded float4 so that it maps onto v_pk_mul_f32 / v_pk_fma_f32 without any lane repacking: left to itself the
// SLP vectoriser paired the same component of DIFFERENT corners, which costs v_mov shuffles right after the loads --
// and a vmcnt wait for them in front of the MFMA block.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 combine(const float4 (&v)[4], const float4& w) {
  v2f lo = v2f{v[0].x, v[0].y} * v2f{w.x, w.x};
  v2f hi = v2f{v[0].z, v[0].w} * v2f{w.x, w.x};
  lo += v2f{v[1].x, v[1].y} * v2f{w.y, w.y};
  hi += v2f{v[1].z, v[1].w} * v2f{w.y, w.y};
  lo += v2f{v[2].x, v[2].y} * v2f{w.z, w.z};
  hi += v2f{v[2].z, v[2].w} * v2f{w.z, w.z};
  lo += v2f{v[3].x, v[3].y} * v2f{w.w, w.w};
  hi += v2f{v[3].z, v[3].w} * v2f{w.w, w.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// ------------------------------------------------------------------------------------------------ forward
// The same kernel serves the INPUT gradient: dx[q][c] = sum_{tap,o} Dg[q][tap][o] * W2t[c][tap*O + o] with
// Dg[q][tap][o] = sum_e w_e dY[p_e][o] gathered through the TRANSPOSED tap table (entries of input pixel q grouped by
// tap: almost always 4, up to 8 in the few rows next to the poles).  `ke` = table entries per (pixel, tap): 4, or 8 with
// `rowmax` (per destination pixel: its largest entry count) telling a tile whether any of its pixels needs the second
// group of four at all (block-uniform; those tiles fetch it at commit time).
// ONE: tables with a single entry per (pixel, tap) (ke == 1) -- an ordinary zero-padded 3x3 convolution seen as a gather
// (the VGG19 stack of the perceptual loss): 4 operand loads per chunk instead of 16, a scale instead of the bilinear
// combine; everything else is the same kernel.
template <int BN, bool ONE = false>
__global__ __launch_bounds__(256, 2) void sphere_conv_fwd_fused_kernel(
    const float* __restrict__ X, const int* __restrict__ idx, const float* __restrict__ wgt,
    const float* __restrict__ W2 /*[O][9C]*/, const float* __restrict__ bias, float* __restrict__ Y /*[M][O]*/, int M,
    int HW /* source pixels per sample */, int Po /* destination pixels per sample */, int C, int O, int ke,
    const unsigned char* __restrict__ rowmax, const float* __restrict__ res /*[M][O] added before the activation, or NULL*/,
    float slope /* leaky-ReLU slope of the epilogue: 1 = none, 0 = ReLU */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][kBM][kLdF]
  float* Bs = smem + 2 * kBM * kLdF;      // [2][BN][kLdF]
  constexpr int NI = BN / 32;             // 16-channel tiles per wave along N
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  With the
  // natural order an XCD would get every 8th 128-pixel tile -- never two vertical neighbours, although the 3 x 3 taps of
  // a tile read the rows above and below it -- and a pixel tile's second O-tile would run a whole grid later: every XCD
  // fetches the whole input from HBM.  Here XCD k walks a CONTIGUOUS band of pixel tiles, the O-tiles of a pixel tile back
  // to back: the 64 tiles in flight on an XCD are ~16 image rows (2 MB at C = 128), which its 4 MB L2 holds, so the
  // gathers of neighbouring taps / tiles / O-tiles are L2 hits.
  const int n_ot = O / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt_local = slot / n_ot, ot = slot - mt_local * n_ot;
  const int mt = xcd * (int)(gridDim.x / (8 * n_ot)) + mt_local;
  if (mt * kBM >= M) return;   // padding of the last band (whole workgroup, before any barrier)
  const int m0 = mt * kBM, o0 = ot * BN;

  // staging roles: A -- pixel sp = tid / 2, 16 channels (half); B -- output channel sp (threads < 2*BN), 16 channels
  const int sp = tid >> 1, half = tid & 1;
  const int ms = min(m0 + sp, M - 1);
  const int sb = ms / Po, spix = ms - sb * Po;
  const float* xb = X + (size_t)sb * HW * C + 16 * half;
  const int* idp = idx + (size_t)spix * 9 * ke;
  const float* wgp = wgt + (size_t)spix * 9 * ke;
  const bool ng2 = (ke == 8) && __syncthreads_or(rowmax ? rowmax[spix] > 4 : 1);   // block-uniform
  const bool stage_b = tid < 2 * BN;
  const float* wrow = W2 + (size_t)(o0 + min(sp, BN - 1)) * 9 * C + 16 * half;

  const int cpt = C / kBK;            // chunks per tap
  const int nchunks = 9 * cpt;

  constexpr int NA = ONE ? 4 : 16;    // gathered float4 per thread and chunk
  constexpr int NP = NA + 4;          // + the dense operand's four
  auto load_tap = [&](int tap) {
    Tap t;
    if constexpr (ONE) {
      t.id = make_int4(idp[tap], -1, -1, -1);
      t.w = make_float4(wgp[tap], 0.f, 0.f, 0.f);
    } else {
      t.id = *reinterpret_cast<const int4*>(idp + ke * tap);
      t.w = *reinterpret_cast<const float4*>(wgp + ke * tap);
    }
    return t;
  };
  float4 av[4][4];          // in-flight operands of the next chunk: A[corner][j] ...
  float4 bv0, bv1, bv2, bv3;  // ... and B (named scalars: as an array the compiler parked them in scratch)
  // the 20 operand loads of a chunk, one at a time: the main loop issues them BETWEEN MFMAs.  Issued back to back
  // in front of the MFMA block they held every wave at the texture-address unit (one 1-KB wave load per ~16 cycles,
  // shared by the CU's 8 waves) for ~2000 cycles per chunk with the matrix pipe idle (measured: 91 -> 139 TF/s when the
  // staging was removed altogether).
  auto load_piece = [&](int piece, int chunk, const Tap& t) {
    const int tap = chunk / cpt, c0 = (chunk - tap * cpt) * kBK;
    if (piece < NA) {
      const int k = piece >> 2, j = piece & 3;
      const int id = k == 0 ? t.id.x : k == 1 ? t.id.y : k == 2 ? t.id.z : t.id.w;
      const float* src = xb + (size_t)max(id, 0) * C + c0;   // out-of-bounds corners carry weight 0
      av[k][j] = *reinterpret_cast<const float4*>(src + 4 * j);
    } else {
      const float* ws = wrow + tap * C + c0 + 4 * (piece - NA);
      const float4 v = *reinterpret_cast<const float4*>(ws);
      if (piece == NA) bv0 = v; else if (piece == NA + 1) bv1 = v; else if (piece == NA + 2) bv2 = v; else bv3 = v;
    }
  };
  auto load_chunk = [&](int chunk, const Tap& t) {
#pragma unroll
    for (int piece = 0; piece < NP; ++piece) load_piece(piece, chunk, t);
  };
  auto gathered = [&](const Tap& t, int j) {
    if constexpr (ONE) {
      const float4 v = av[0][j];
      return make_float4(v.x * t.w.x, v.y * t.w.x, v.z * t.w.x, v.w * t.w.x);
    } else {
      const float4 v[4] = {av[0][j], av[1][j], av[2][j], av[3][j]};
      return combine(v, t.w);
    }
  };
  // commit = bilinear combine + LDS store of the staged operands.  Tiles without pole rows (ng2 == false, 97 % of
  // them) commit piecewise from INSIDE the second K-half's MFMA stream (commit_a / commit_b); the rare ng2 tiles
  // commit after it, fetching slots 4..7 of the transposed table with the latency exposed.
  auto commit_a = [&](int buf, const Tap& t, int j) {
    float* ad = As + (size_t)buf * kBM * kLdF + sp * kLdF + 16 * half;
    *reinterpret_cast<float4*>(ad + 4 * j) = gathered(t, j);
  };
  auto commit_b = [&](int buf, int j) {
    if (BN == 128 || stage_b) {   // BN = 128: every thread stages a weight row (no branch in the MFMA stream)
      float* bd = Bs + (size_t)buf * BN * kLdF + sp * kLdF + 16 * half;
      *reinterpret_cast<float4*>(bd + 4 * j) = j == 0 ? bv0 : j == 1 ? bv1 : j == 2 ? bv2 : bv3;
    }
  };
  auto commit_chunk = [&](int buf, const Tap& t, int chunk) {
    float* ad = As + (size_t)buf * kBM * kLdF + sp * kLdF + 16 * half;
    float4 out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = gathered(t, j);
    if (!ONE && ng2) {   // entries 4..7 of this (pixel, tap)
      const int tap = chunk / cpt, c0 = (chunk - tap * cpt) * kBK;
      const int4 id2 = *reinterpret_cast<const int4*>(idp + ke * tap + 4);
      const float4 w2 = *reinterpret_cast<const float4*>(wgp + ke * tap + 4);
      const int ids[4] = {id2.x, id2.y, id2.z, id2.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* src = xb + (size_t)max(ids[k], 0) * C + c0;
#pragma unroll
        for (int j = 0; j < 4; ++j) av[k][j] = *reinterpret_cast<const float4*>(src + 4 * j);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v[4] = {av[0][j], av[1][j], av[2][j], av[3][j]};
        const float4 e = combine(v, w2);
        out[j].x += e.x;
        out[j].y += e.y;
        out[j].z += e.z;
        out[j].w += e.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(ad + 4 * j) = out[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) commit_b(buf, j);
  };

  f32x4 acc[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // t_use: table entry of the tap whose chunk is being loaded / committed; t_pref: the next tap's entry, requested a
  // whole tap (>= one chunk of MFMAs) before it is needed
  Tap t_use = load_tap(0);
  Tap t_pref = load_tap(1);
  load_chunk(0, t_use);
  commit_chunk(0, t_use, 0);
  __syncthreads();
  auto chunk_loop = [&](auto ng2_c) {
    constexpr bool NG2 = decltype(ng2_c)::value;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const int buf = chunk & 1;
      const int nxt = chunk + 1;
      const bool has_next = nxt < nchunks;
      if (has_next && nxt % cpt == 0) {   // block-uniform: chunk nxt opens a new tap
        t_use = t_pref;
        t_pref = load_tap(min(nxt / cpt + 1, 8));
      }
      const int nxc = has_next ? nxt : chunk;   // the last chunk re-requests its own operands (committed to the idle buffer)
      __builtin_amdgcn_sched_barrier(0);
      const float* ab = As + (size_t)buf * kBM * kLdF + (64 * wm + r) * kLdF + 8 * kk;
      const float* bb = Bs + (size_t)buf * BN * kLdF + ((BN / 2) * wn + r) * kLdF + 8 * kk;
      constexpr int kHalf = 4 * NI * 4;       // MFMAs per K-half: 64 (BN = 128) or 32 (BN = 64)
      // first half: one operand load per kEvery MFMAs; second half: the 8 commit pieces from its middle on.  (Tried:
      // all loads within the first 2/3 of the half and commits only in the last quarter -- no change, 109 vs 112 TF/s.)
      constexpr int kEvery = kHalf / NP;
      constexpr int kCommit0 = kHalf / 2;
      constexpr int kCommitEvery = kHalf / 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 af[4], bf[NI];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const float4*>(ab + 16 * mi * kLdF + 4 * h);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const float4*>(bb + 16 * ni * kLdF + 4 * h);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              acc[ni][mi] = mfma16(f4c(bf[ni], t), f4c(af[mi], t), acc[ni][mi]);
              const int cnt = (t * NI + ni) * 4 + mi;          // MFMA index within this K-half
              if (h == 0 && cnt % kEvery == kEvery - 1 && cnt / kEvery < NP) {
                load_piece(cnt / kEvery, nxc, t_use);
                __builtin_amdgcn_sched_barrier(0);             // keep the request here (the scheduler would sink it)
              }
              if (!NG2 && h == 1 && cnt >= kCommit0 && (cnt - kCommit0) % kCommitEvery == kCommitEvery - 1) {
                const int piece = (cnt - kCommit0) / kCommitEvery;   // 0..7: four A quads, then four B quads
                if (piece < 4) commit_a(buf ^ 1, t_use, piece); else commit_b(buf ^ 1, piece - 4);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
      }
      __builtin_amdgcn_sched_barrier(0);
      // unconditional (the last chunk commits its own operands again into the idle buffer): under `if (has_next)` the
      // loads' only use sat in a successor block and LLVM sank all 20 of them there, behind the MFMAs
      if constexpr (NG2) commit_chunk(buf ^ 1, t_use, nxc);
      eml::lds_barrier();
    }
  };
  if (ng2) chunk_loop(std::true_type{}); else chunk_loop(std::false_type{});
  // epilogue: lane (r, kk) owns output channels 4kk..4kk+3 of tile ni for pixel r of tile mi
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int o = o0 + (BN / 2) * wn + 16 * ni + 4 * kk;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bq = *reinterpret_cast<const float4*>(bias + o);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + 64 * wm + 16 * mi + r;
      if (m < M) {
        float4 v = make_float4(acc[ni][mi][0] + bq.x, acc[ni][mi][1] + bq.y, acc[ni][mi][2] + bq.z, acc[ni][mi][3] + bq.w);
        if (res) {   // residual sum of a ResNet block (x_s + conv_1(.)): the lane that writes an element reads it, Y may alias res
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

// ------------------------------------------------------------------------------------------------ weight gradient
// grid = (column tiles of the (tap, c) axis, row tiles of O, split-K); partial[z][O][9C].  BMO = output channels per
// tile: 128, or 64 for the 64-channel layers (a 128-row tile would spend half of its MFMAs on padding).
// VR (round 4): MFMA row / column r of tile j stands for channel MI*r + j (resp. NI*r + j) of the wave's span instead of
// 16*j + r -- which channel a row index means is free as long as the epilogue agrees -- so a lane's MI (NI) operand values of a
// k-step are CONSECUTIVE floats of one LDS row: one ds_read_b128 (b64) instead of four (two) ds_read_b32, 16 LDS read
// instructions per chunk and wave instead of 64, and the result leaves as 16-byte stores.  Layout and commits are unchanged.
template <int BN, int BMO, bool VR>
__global__ __launch_bounds__(256, 2) void sphere_conv_wgrad_fused_kernel(
    const float* __restrict__ X, const int* __restrict__ idx, const float* __restrict__ wgt,
    const float* __restrict__ dY /*[M][O]*/, float* __restrict__ partial, int M, int HW, int Po, int C, int O,
    int xcd_group) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;                       // [2][kBK][kLdW]  dY tile   [pixel][o]
  float* Gs = smem + 2 * kBK * kLdW;      // [2][kBK][kLdW]  Ag tile   [pixel][c]
  constexpr int NI = BN / 32;
  constexpr int MI = BMO / 32;            // 16-row tiles per wave along O
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_per_tap = C / BN;
  // (natural 3-D grid order: sending all (tap, c-tile, O-tile) workgroups of a K-split to one XCD, the forward
  // kernel's trick, made this kernel 10 % slower -- measured)
  // xcd_group (the launcher decides): workgroups are dealt round-robin to the 8 XCDs by their linear id; all (tap, c-tile, O-tile)
  // tiles of a K-split then run on ONE XCD (splits z = k, k + 8, ... on XCD k; gridDim.z is a multiple of 8): the dY chunk
  // and the source rows the 9 taps share are fetched into one L2 instead of eight
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (xcd_group) {
    const int T = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3;
    const int zq = slot / T, t = slot - zq * T;
    bz = 8 * zq + xcd;
    by = t / (int)gridDim.x;
    bx = t - by * (int)gridDim.x;
  }
  const int tap = bx / tiles_per_tap, c0 = (bx - tap * tiles_per_tap) * BN;
  const int o0 = by * BMO;
  const int S = gridDim.z;
  const int nchunks_all = (M + kBK - 1) / kBK;

  // staging roles: pixel sp = tid / 8 of the chunk; 16 consecutive columns starting at 16 * (tid % 8)
  const int sp = tid >> 3, sq = tid & 7;
  const bool stage_g = 16 * sq < BN;      // BN = 64: half of the threads gather
  const bool stage_d = 16 * sq < BMO;     // BMO = 64: half of the threads stage dY
  const int ocol = min(o0 + 16 * sq, O - 16);   // O >= 64 and a multiple of 16: clamped columns are zeroed at commit

  float4 dv[4], gv[4][4];
  float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
  bool pvalid = false;
  auto tap_of = [&](int chunk, Tap& t, size_t& xoff) {
    const int m = min(chunk * kBK + sp, M - 1);
    const int b = m / Po, p = m - b * Po;
    t.id = *reinterpret_cast<const int4*>(idx + ((size_t)p * 9 + tap) * 4);
    t.w = *reinterpret_cast<const float4*>(wgt + ((size_t)p * 9 + tap) * 4);
    xoff = (size_t)b * HW * C;
  };
  // the 20 operand loads of a chunk one at a time (pieces 0..3: dY, 4..19: the gathered corners) and its commit in 8
  // pieces: the main loop issues them BETWEEN MFMAs (see the forward kernel)
  const float* dsrc_n = dY;
  auto begin_chunk = [&](int chunk, const Tap& t) {
    const int m = chunk * kBK + sp;
    pvalid = m < M;
    dsrc_n = dY + (size_t)min(m, M - 1) * O + ocol;
    gw = t.w;
  };
  auto load_piece = [&](int piece, const Tap& t, size_t xoff) {
    if (piece < 4) {
      dv[piece] = *reinterpret_cast<const float4*>(dsrc_n + 4 * piece);
    } else {
      const int k = (piece - 4) >> 2, j = (piece - 4) & 3;
      const int id = k == 0 ? t.id.x : k == 1 ? t.id.y : k == 2 ? t.id.z : t.id.w;
      const float* src = X + xoff + (size_t)max(id, 0) * C + c0 + (stage_g ? 16 * sq : 0);
      gv[k][j] = *reinterpret_cast<const float4*>(src + 4 * j);
    }
  };
  auto load_chunk = [&](int chunk, const Tap& t, size_t xoff) {
    begin_chunk(chunk, t);
#pragma unroll
    for (int piece = 0; piece < 20; ++piece) load_piece(piece, t, xoff);
  };
  auto commit_piece = [&](int buf, int piece) {   // 0..3: dY quads, 4..7: gathered quads
    if (piece < 4) {
      float* dd = Ds + (size_t)buf * kBK * kLdW + sp * kLdW + 16 * sq;
      const bool ov = o0 + 16 * sq + 16 <= O;
      float4 v = dv[piece];
      if (!(pvalid && ov)) v = make_float4(0.f, 0.f, 0.f, 0.f);   // pixels past M / columns past O add 0
      if (stage_d) *reinterpret_cast<float4*>(dd + 4 * piece) = v;
    } else if (stage_g) {
      const int j = piece - 4;
      float* gd = Gs + (size_t)buf * kBK * kLdW + sp * kLdW + 16 * sq;
      const float4 v[4] = {gv[0][j], gv[1][j], gv[2][j], gv[3][j]};
      *reinterpret_cast<float4*>(gd + 4 * j) = combine(v, gw);
    }
  };
  auto commit_chunk = [&](int buf) {
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) commit_piece(buf, piece);
  };

  f32x4 acc[MI][NI];   // [o tile][c tile]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  int chunk = bz;
  Tap t0, t1;
  size_t x0 = 0, x1 = 0;
  if (chunk < nchunks_all) {
    tap_of(chunk, t0, x0);
    load_chunk(chunk, t0, x0);
    commit_chunk(0);
    tap_of(min(chunk + S, nchunks_all - 1), t1, x1);
  }
  __syncthreads();
  int it = 0;
  constexpr int kSteps = kBK / 4;                  // 8 k-steps of MI*NI MFMAs
  constexpr int kPerStep = MI * NI;
  constexpr int kLoadEvery = (kPerStep * 4 >= 20) ? (kPerStep * 4) / 20 : 1;   // loads ride on the first 4 k-steps
  for (; chunk < nchunks_all; chunk += S, ++it) {
    const int buf = it & 1;
    // the last chunk re-requests its own operands and commits them into the idle buffer: an `if (has_next)` around the
    // commit would let LLVM sink the loads into that branch, behind the MFMAs
    const int nxt = min(chunk + S, nchunks_all - 1);
    const Tap tn = t1;
    const size_t xn = x1;
    begin_chunk(nxt, tn);
    __builtin_amdgcn_sched_barrier(0);
    const float* db = Ds + (size_t)buf * kBK * kLdW + (BMO / 2) * wm + (VR ? MI * r : r);
    const float* gb = Gs + (size_t)buf * kBK * kLdW + (BN / 2) * wn + (VR ? NI * r : r);
#pragma unroll
    for (int ks = 0; ks < kSteps; ++ks) {
      float a[MI], b[NI];
      if constexpr (VR) {
        typedef float vmi __attribute__((ext_vector_type(MI)));
        typedef float vni __attribute__((ext_vector_type(NI)));
        const vmi av = *reinterpret_cast<const vmi*>(db + (4 * ks + kk) * kLdW);
        const vni bv = *reinterpret_cast<const vni*>(gb + (4 * ks + kk) * kLdW);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = av[mi];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = bv[ni];
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = db[(4 * ks + kk) * kLdW + 16 * mi];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = gb[(4 * ks + kk) * kLdW + 16 * ni];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = mfma16(a[mi], b[ni], acc[mi][ni]);
          const int cnt = ks * kPerStep + mi * NI + ni;      // MFMA index within the chunk
          if (cnt % kLoadEvery == kLoadEvery - 1 && cnt / kLoadEvery < 20) {
            load_piece(cnt / kLoadEvery, tn, xn);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (cnt == kLoadEvery * 20) {                      // the table entry for the chunk after that
            tap_of(min(nxt + S, nchunks_all - 1), t1, x1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (ks >= kSteps - 4) {                                // commits ride on the last 4 k-steps, two pieces each
        commit_piece(buf ^ 1, 2 * (ks - (kSteps - 4)));
        commit_piece(buf ^ 1, 2 * (ks - (kSteps - 4)) + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    eml::lds_barrier();
  }
  // partial[z][o][tap*C + c]: lane (r, kk) holds rows o = 4kk + g, column c = r of each 16x16 tile
  float* out = partial + (size_t)bz * O * 9 * C;
  if constexpr (VR) {
    // tile (mi, ni): row index i = 4kk + g is output channel MI*i + mi, column index r is input channel NI*r + ni
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int o = o0 + (BMO / 2) * wm + MI * (4 * kk + g) + mi;
        if (o < O) {
          typedef float vni __attribute__((ext_vector_type(NI)));
          vni v;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) v[ni] = acc[mi][ni][g];
          *reinterpret_cast<vni*>(out + (size_t)o * 9 * C + tap * C + c0 + (BN / 2) * wn + NI * r) = v;
        }
      }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int o = o0 + (BMO / 2) * wm + 16 * mi + 4 * kk + g;
      if (o < O) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          out[(size_t)o * 9 * C + tap * C + c0 + (BN / 2) * wn + 16 * ni + r] = acc[mi][ni][g];
      }
    }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int S, size_t n,
                                                           float* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += partial[(size_t)s * n + e];
  out[e] = t;
}

}  // namespace

namespace {
// EML_GG_V1=1: A/B switch back to the round-2 kernel for every table (read once)
bool gg_v1_enabled() {
  static const bool on = [] { const char* v = getenv("EML_GG_V1"); return v && v[0] == '1'; }();
  return on;
}
// gather_gemm2's per-pixel 32-bit BYTE offsets are relative to the tile's first sample: a 128-pixel tile spans at most
// (127 / Po + 2) samples of HW * C floats each (the HW * C bound alone does not cover Po < 128)
bool gg2_offsets_fit(long HW, int C, int Po) {
  return (unsigned long long)HW * C < (1ull << 29) &&
         (unsigned long long)((gg2::kBM - 1) / Po + 2) * (unsigned long long)HW * C * 4ull < (1ull << 32);
}
// EML_GG_NOSHARE=1: A/B switch -- ignore EML_TAP_ROWSHARE (the 16-load gather for every table)
bool gg_noshare() {
  static const bool on = [] { const char* v = getenv("EML_GG_NOSHARE"); return v && v[0] == '1'; }();
  return on;
}
int launch_gather_gemm(const char* what, const float* X, const int* idx, const float* wgt, const float* W2, const float* bias,
                       float* Y, int B, int HW, int Po, int C, int O, int ke, const unsigned char* rowmax,
                       const float* res, float slope, eml_stream_t stream, int table_flags = 0) {
  if (B == 0) return EML_OK;
  const long M = (long)B * Po;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "%s: too many pixels", what);
  const int bn = (O % 128 == 0) ? 128 : 64;
  // Round 4: the second-generation kernel (gather_gemm2.h: dense operand and tap table by LDS-DMA, half-line gathers, one
  // 32-bit offset per load, pole rows as a second virtual tap) for the bilinear tables -- +2 ... +15 % per layer shape on
  // the same box (profiles/r04_gg2_variants.jsonl, variant v2.2).  It needs two chunks per tap (C >= 64) and its 32-bit
  // offsets to reach two samples / the weight tile; single-entry (planar) tables keep the round-2 kernel, which schedules
  // their four loads per chunk better (vgg 64 -> 64: 107 against 98 TF/s).
  if (ke != 1 && C >= 64 && gg2_offsets_fit(HW, C, Po) && (unsigned long long)O * 9 * C < (1ull << 30) && !gg_v1_enabled()) {
    const long n_mt2 = (M + gg2::kBM - 1) / gg2::kBM, per_xcd2 = (n_mt2 + 7) / 8;
    const dim3 grid2((unsigned)(8 * per_xcd2 * (O / bn)));
#define EML_LAUNCH_GG2(BNV, LPPV, SHV)                                                                               \
  do {                                                                                                              \
    auto kern = gg2::gather_gemm2_kernel<BNV, 256, LPPV, false, false, false, SHV>;                                 \
    EML_ENSURE_LDS(kern, (gg2::lds_bytes<BNV, 256>()));                                                             \
    hipLaunchKernelGGL(kern, grid2, dim3(256), (gg2::lds_bytes<BNV, 256>()), (hipStream_t)stream, X, idx, wgt, W2,  \
                       bias, Y, (int)M, HW, Po, C, O, ke, rowmax, res, slope, gg2::SpadeEpilogue{});                \
  } while (0)
    // round 5: row-shared corners (10 gathered loads per chunk instead of 16) where the caller vouches for the table
    const bool share = (table_flags & EML_TAP_ROWSHARE) && ke == 4 && Po % 4 == 0 && !gg_noshare();
    if (share) {
      if (bn == 128) EML_LAUNCH_GG2(128, 8, true); else EML_LAUNCH_GG2(64, 8, true);
    } else {
      if (bn == 128) EML_LAUNCH_GG2(128, 4, false); else EML_LAUNCH_GG2(64, 4, false);
    }
#undef EML_LAUNCH_GG2
    return eml::check_launch(what);
  }
  const size_t lds = (size_t)(2 * kBM * kLdF + 2 * bn * kLdF) * sizeof(float);
  const long n_mt = (M + kBM - 1) / kBM, per_xcd = (n_mt + 7) / 8;
  const dim3 grid((unsigned)(8 * per_xcd * (O / bn)));   // 1-D: the kernel maps id -> (XCD band, pixel tile, O-tile)
#define EML_LAUNCH_GG(BNV, ONEV)                                                                                     \
  do {                                                                                                              \
    EML_ENSURE_LDS((&sphere_conv_fwd_fused_kernel<BNV, ONEV>), lds);                                                \
    hipLaunchKernelGGL((sphere_conv_fwd_fused_kernel<BNV, ONEV>), grid, dim3(256), lds, (hipStream_t)stream, X, idx, \
                       wgt, W2, bias, Y, (int)M, HW, Po, C, O, ke, rowmax, res, slope);                             \
  } while (0)
  if (ke == 1) {
    if (bn == 128) EML_LAUNCH_GG(128, true); else EML_LAUNCH_GG(64, true);
  } else {
    if (bn == 128) EML_LAUNCH_GG(128, false); else EML_LAUNCH_GG(64, false);
  }
#undef EML_LAUNCH_GG
  return eml::check_launch(what);
}
}  // namespace

extern "C" int eml_sphere_conv_fwd_fused_f32(const float* X, const int* idx, const float* wgt, const float* W2,
                                             const float* bias, float* Y, int B, int HW, int Po, int C, int O,
                                             int ke, eml_stream_t stream) {
  if (!X || !idx || !wgt || !W2 || !Y || B < 0 || HW < 1 || Po < 1 || C < 32 || (C % 32) || O < 64 || (O % 64))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_fwd_fused_f32: need C %% 32 == 0, O %% 64 == 0 (C=%d, O=%d)", C, O);
  if (ke != 4 && ke != 1) return eml::fail(EML_EINVAL, "eml_sphere_conv_fwd_fused_f32: ke must be 4 (bilinear taps) or 1");
  return launch_gather_gemm("eml_sphere_conv_fwd_fused_f32", X, idx, wgt, W2, bias, Y, B, HW, Po, C, O, ke, nullptr, nullptr,
                            1.f, stream);
}

// The same product with the epilogue of its consumer folded in: Y = act(conv + bias + residual), act = leaky ReLU of slope
// `act_slope` (1 = none, 0 = ReLU; in [0, 1]).  `residual` (B*Po, O) or NULL; Y may alias it.
extern "C" int eml_sphere_conv_fwd_fused_ex_f32(const float* X, const int* idx, const float* wgt, const float* W2,
                                                const float* bias, float* Y, int B, int HW, int Po, int C, int O, int ke,
                                                const float* residual, float act_slope, int table_flags,
                                                eml_stream_t stream) {
  if (!X || !idx || !wgt || !W2 || !Y || B < 0 || HW < 1 || Po < 1 || C < 32 || (C % 32) || O < 64 || (O % 64))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_fwd_fused_ex_f32: need C %% 32 == 0, O %% 64 == 0 (C=%d, O=%d)", C, O);
  if (ke != 4 && ke != 1) return eml::fail(EML_EINVAL, "eml_sphere_conv_fwd_fused_ex_f32: ke must be 4 (bilinear taps) or 1");
  if (!(act_slope >= 0.f && act_slope <= 1.f))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_fwd_fused_ex_f32: act_slope %g outside [0, 1]", (double)act_slope);
  if (table_flags & ~EML_TAP_ROWSHARE) return eml::fail(EML_EINVAL, "eml_sphere_conv_fwd_fused_ex_f32: unknown table_flags %d", table_flags);
  return launch_gather_gemm("eml_sphere_conv_fwd_fused_ex_f32", X, idx, wgt, W2, bias, Y, B, HW, Po, C, O, ke, nullptr,
                            residual, act_slope, stream, table_flags);
}

// SPADE (normalization.py:101-115) in one launch: the gamma | beta SphereConv (128 -> 2 Cn over `actv`) with the modulation of
// the parameter-free-normalised map x as its epilogue,  Y = leaky_relu(((x - mean) * istd) * (1 + gamma) + beta, slope).
// W2r (2 Cn, 9 Cin) / bias_r (2 Cn): the rows of cat(gamma head, beta head) in the kernel's order -- position p holds
// source row  c + half * Cn  with  c = 64 (p / 128) + 32 ((p % 128) / 64) + p % 32,  half = (p % 64) / 32.
extern "C" int eml_sphere_conv_spade_supported(int Cin, int Cn, long HW) {
  return Cin >= 64 && Cin % 32 == 0 && Cn >= 64 && Cn % 64 == 0 && HW >= 1 && HW <= 2147483647L &&
         gg2_offsets_fit(HW, Cin, (int)HW) && (unsigned long long)2 * Cn * 9 * Cin < (1ull << 30);
}

extern "C" int eml_sphere_conv_spade_fwd_f32(const float* actv, const int* idx, const float* wgt, const float* W2r,
                                             const float* bias_r, const float* x, const float* mean, const float* istd,
                                             float* Y, float* gamma_out, int B, int H, int W, int Cin, int Cn, int up2,
                                             float act_slope, int table_flags, eml_stream_t stream) {
  if (!actv || !idx || !wgt || !W2r || !x || !mean || !istd || !Y || B < 0 || H < 1 || W < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_spade_fwd_f32: null pointer / empty grid");
  if (!eml_sphere_conv_spade_supported(Cin, Cn, (long)H * W))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_spade_fwd_f32: unsupported widths (Cin=%d, Cn=%d, HW=%ld)", Cin, Cn, (long)H * W);
  if (up2 && ((H | W) & 1)) return eml::fail(EML_EINVAL, "eml_sphere_conv_spade_fwd_f32: up2 needs an even grid (%d x %d)", H, W);
  if (!(act_slope >= 0.f && act_slope <= 1.f))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_spade_fwd_f32: act_slope %g outside [0, 1]", (double)act_slope);
  if (B == 0) return EML_OK;
  const int HW = H * W, O = 2 * Cn;
  const long M = (long)B * HW;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_spade_fwd_f32: too many pixels");
  const long n_mt = (M + gg2::kBM - 1) / gg2::kBM, per_xcd = (n_mt + 7) / 8;
  const dim3 grid((unsigned)(8 * per_xcd * (O / 128)));
  if (table_flags & ~EML_TAP_ROWSHARE) return eml::fail(EML_EINVAL, "eml_sphere_conv_spade_fwd_f32: unknown table_flags %d", table_flags);
  const gg2::SpadeEpilogue mod{x, mean, istd, gamma_out, up2 ? 1 : 0, H, W};
  if ((table_flags & EML_TAP_ROWSHARE) && HW % 4 == 0 && !gg_noshare()) {
    auto kern = gg2::gather_gemm2_kernel<128, 256, 8, false, false, true, true>;
    EML_ENSURE_LDS(kern, (gg2::lds_bytes<128, 256>()));
    hipLaunchKernelGGL(kern, grid, dim3(256), (gg2::lds_bytes<128, 256>()), (hipStream_t)stream, actv, idx, wgt, W2r, bias_r, Y,
                       (int)M, HW, HW, Cin, O, 4, nullptr, nullptr, act_slope, mod);
  } else {
    auto kern = gg2::gather_gemm2_kernel<128, 256, 4, false, false, true>;
    EML_ENSURE_LDS(kern, (gg2::lds_bytes<128, 256>()));
    hipLaunchKernelGGL(kern, grid, dim3(256), (gg2::lds_bytes<128, 256>()), (hipStream_t)stream, actv, idx, wgt, W2r, bias_r, Y,
                       (int)M, HW, HW, Cin, O, 4, nullptr, nullptr, act_slope, mod);
  }
  return eml::check_launch("eml_sphere_conv_spade_fwd_f32");
}

// dX (B*HW, C) = gather-GEMM over the transposed tap table: tidx / twgt (HW*9*ke) = for input pixel q and tap t the
// output pixels whose tap t samples q (-1 = empty slot, weight 0) and their bilinear weights; W2t (C, 9*O), columns (tap, o)
extern "C" int eml_sphere_conv_dgrad_fused_f32(const float* dY, const int* tidx, const float* twgt,
                                               const unsigned char* rowmax, int ke, const float* W2t, float* dX, int B,
                                               int HW, int Po, int C, int O, int table_flags, eml_stream_t stream) {
  if (!dY || !tidx || !twgt || !W2t || !dX || B < 0 || HW < 1 || Po < 1 || O < 32 || (O % 32) || C < 64 || (C % 64) ||
      (ke != 4 && ke != 8 && ke != 1) || (ke == 8 && !rowmax))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_dgrad_fused_f32: need O %% 32 == 0, C %% 64 == 0, ke in {1, 4, 8} (C=%d, O=%d)",
                     C, O);
  // roles swap: the rows gathered are dY's (Po per sample, O wide), the destination pixels are the HW input pixels
  if (table_flags & ~EML_TAP_ROWSHARE) return eml::fail(EML_EINVAL, "eml_sphere_conv_dgrad_fused_f32: unknown table_flags %d", table_flags);
  return launch_gather_gemm("eml_sphere_conv_dgrad_fused_f32", dY, tidx, twgt, W2t, nullptr, dX, B, Po, HW, O, C, ke, rowmax,
                            nullptr, 1.f, stream, table_flags);
}

extern "C" size_t eml_sphere_conv_wgrad_partial_floats(int C, int O, int split_k) {
  return (size_t)split_k * O * 9 * C;
}

extern "C" int eml_sphere_conv_wgrad_fused_f32(const float* X, const int* idx, const float* wgt, const float* dY,
                                               float* partial, float* dW2, int B, int HW, int Po, int C, int O,
                                               int split_k, eml_stream_t stream) {
  if (!X || !idx || !wgt || !dY || !partial || !dW2 || B < 1 || HW < 1 || Po < 1 || C < 64 || (C % 64) || O < 64 ||
      (O % 16) || split_k < 1 || split_k > 65535)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_wgrad_fused_f32: need C %% 64 == 0, O >= 64, O %% 16 == 0 (C=%d, O=%d)", C, O);
  const long M = (long)B * Po;
  if (M > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_wgrad_fused_f32: too many pixels");
  const int bn = (C % 128 == 0) ? 128 : 64;
  const int bmo = (O % 128 == 0 || O > 192) ? 128 : 64;   // 64-wide row tiles when a 128-row tile would be mostly padding
  const size_t lds = (size_t)(4 * kBK * kLdW) * sizeof(float);
  // The tiles of a K-split on one XCD (see the kernel) where a split has at most 18 of them -- then the 7 splits an XCD
  // holds at a time are all resident: +1.6 ... 2.6 % on the 128 x 256 layers, +-0 at 64 x 128; with 36 / 72 tiles per split
  // (128 -> 512, 512 -> 256) the same order LOSES 12 / 22 % (profiles/r06_wgrad_xcd.txt).  Needs a multiple of 8 splits: the
  // remainder of the caller's split_k is not used (partial rows past it are never summed).  EML_WG_XCD=0: natural order (A/B).
  static const bool xcd_env = [] { const char* v = getenv("EML_WG_XCD"); return !(v && v[0] == '0'); }();
  const int xg = (xcd_env && split_k >= 8 && 9 * (C / bn) * ((O + bmo - 1) / bmo) <= 18) ? 1 : 0;
  if (xg) split_k &= ~7;
  const dim3 grid(9 * (C / bn), (O + bmo - 1) / bmo, split_k);
  static const bool scalar_reads = [] { const char* v = getenv("EML_WG_V1"); return v && v[0] == '1'; }();   // A/B switch
#define EML_LAUNCH_WGRAD(BNV, BMV)                                                                                   \
  do {                                                                                                              \
    if (scalar_reads) {                                                                                             \
      EML_ENSURE_LDS((&sphere_conv_wgrad_fused_kernel<BNV, BMV, false>), lds);                                      \
      hipLaunchKernelGGL((sphere_conv_wgrad_fused_kernel<BNV, BMV, false>), grid, dim3(256), lds, (hipStream_t)stream, X, idx, \
                         wgt, dY, partial, (int)M, HW, Po, C, O, xg);                                                \
    } else {                                                                                                        \
      EML_ENSURE_LDS((&sphere_conv_wgrad_fused_kernel<BNV, BMV, true>), lds);                                       \
      hipLaunchKernelGGL((sphere_conv_wgrad_fused_kernel<BNV, BMV, true>), grid, dim3(256), lds, (hipStream_t)stream, X, idx, \
                         wgt, dY, partial, (int)M, HW, Po, C, O, xg);                                                \
    }                                                                                                               \
  } while (0)
  if (bn == 128) {
    if (bmo == 128) EML_LAUNCH_WGRAD(128, 128); else EML_LAUNCH_WGRAD(128, 64);
  } else {
    if (bmo == 128) EML_LAUNCH_WGRAD(64, 128); else EML_LAUNCH_WGRAD(64, 64);
  }
#undef EML_LAUNCH_WGRAD
  int rc = eml::check_launch("eml_sphere_conv_wgrad_fused_f32");
  if (rc) return rc;
  const size_t n = (size_t)O * 9 * C;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial,
                     split_k, n, dW2);
  return eml::check_launch("eml_sphere_conv_wgrad_fused_f32(reduce)");
}
