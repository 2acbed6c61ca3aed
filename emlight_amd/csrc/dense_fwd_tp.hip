// conv3x3 of a dense layer, TAP-PACKED (round 6): BN2 -> conv2 (48 -> 12, DenseNet.py:38-43) with no padded MFMA rows.
//
// conv3x3_fwd_kernel (dense_fwd.hip) computes out[o][p] = sum_{tap,c} W[o][tap,c] zn[p + tap][c] with the 12 output channels
// as the rows of v_mfma_f32_16x16x4_f32: 12 of 16 rows carry work (the 0.75 ceiling of DESIGN 3.1), 108 MFMAs per 16 pixels,
// and the B operand is a 10x34x48 halo tile staged through LDS and read back 9 times.  Here the product is turned round:
//
//     D[(tap, o)][q] = sum_c W[o][c][tap] * zn[q][c]           one INPUT pixel q, all 9 taps x 12 channels = 108 rows
//     out[o][p]      = sum_tap D[(tap, o)][p + tap]             a shift-and-add of accumulator registers
//
// 108 rows fill 7 row tiles (112 rows, 96 %): K = 48 channels = 12 MFMA steps x 7 tiles = 84 MFMAs per 16 pixels instead of
// 108 (-22 %).  Every input pixel is a B operand exactly once, so z goes HBM -> registers (BN2 applied there) -> MFMA like
// the 1x1 kernel's operand: no halo tile, no LDS staging, no fragment re-reads.  The rows are ordered so that the shift-and-
// add never crosses 16-lane rows: lane group kk of the accumulator (rows 4kk..4kk+3 of each tile: 28 slots) holds the 9 taps
// of output channels 3kk..3kk+2 (slot s = 3 tap + o % 3, tile s / 4, register s % 4), so
//   * dy moves a contribution to another output ROW: a workgroup walks down a band of image rows and keeps a rolling window
//     of three output rows in registers (rows y-1, y, y+1 while input row y is multiplied); row y-1 is complete and leaves
//     after input row y;
//   * dx moves it to the neighbouring pixel = the neighbouring lane: v_add_f32 with DPP row_shr:1 / row_shl:1, the lane that
//     falls off a 16-pixel tile goes to the next tile's window register (row_shl:15 / row_shr:15 with zero fill);
//   * a wave owns TX tiles of a row, NW waves side by side cover the whole image width (W = 16 TX NW), and the two edge
//     pixels of a wave hand 9 values per lane group to the neighbour wave through LDS -- one LDS-only barrier per row.
// Bands: a workgroup owns `band` output rows of one image; the input rows just above / below the band are multiplied with
// the 3 row tiles that hold their dy = -1 / dy = +1 taps only (3 of 7 tiles: 0.86 rows of overhead per band).
// Sums are formed per tap over the 48 channels and then over the 9 taps in a fixed order: deterministic, f32 round-off of
// the other kernel's single 432-term accumulation.  Batch statistics of the 12 new channels as in conv3x3_fwd_kernel.
#include <type_traits>

#include "eml_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; }

constexpr int kNWMax = 4;

// DPP row shifts with zero fill (a DPP row = the 16 pixels of a tile):  shr<n>: lane x <- lane x - n;  shl<n>: lane x <- x + n
__device__ __forceinline__ float shr1(float v) { return eml::dpp_mov<0x111>(v); }
__device__ __forceinline__ float shl1(float v) { return eml::dpp_mov<0x101>(v); }
__device__ __forceinline__ float shr15(float v) { return eml::dpp_mov<0x11F>(v); }   // lane 15 <- lane 0, zeros elsewhere
__device__ __forceinline__ float shl15(float v) { return eml::dpp_mov<0x10F>(v); }   // lane 0 <- lane 15, zeros elsewhere

// which row tiles a row is multiplied with
enum { kFull = 0, kTop = 1 /* input row above the band: dy = -1 taps, tiles 0..2 */, kBot = 2 /* below: dy = +1, tiles 4..6 */ };

template <int TX>
__global__ __launch_bounds__(256, 2) void conv3x3_fwd_tp_kernel(
    const float* __restrict__ Z, const float* __restrict__ scale2, const float* __restrict__ shift2,
    const float* __restrict__ W2t /* [7][3][64] float4 */, float* __restrict__ X, int ldx, int c_out0, int B, int H, int W,
    int band, double* __restrict__ partials) {
  __shared__ float junk[64 + 36];
  __shared__ float edge[2][kNWMax][2][9][4];   // [row parity][wave][side][3 (dy + 1) + o][lane group]: written by the wave's edge lanes
  __shared__ double red[kNWMax][12][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = blockDim.x >> 6;
  const int x = lane & 15, kk = lane >> 4;

  // weight fragments: lane (row i, k index kq) of tile `tile`, 16-channel group J: W2[o][16J + 4kq + t][tap], t = 0..3
  // row tiles 0, 1 in registers (24), tiles 2..6 in LDS (15 KB, one ds_read_b128 per fragment and 16-channel group): all 84
  // registers + the three window rows + the accumulators do not fit the 256 registers of two waves per SIMD (measured with
  // 2 / 3 / 4 / 5 register-resident tiles: 0.618 / 0.626 / 0.627 / 0.648 ms per layer, profiles/r06_c3tp_variants.txt)
#ifndef TP_KTR
#define TP_KTR 2
#endif
  constexpr int kTR = TP_KTR;
  float4 wf[kTR > 0 ? kTR : 1][3];
#pragma unroll
  for (int tile = 0; tile < kTR; ++tile)
#pragma unroll
    for (int J = 0; J < 3; ++J) wf[tile][J] = reinterpret_cast<const float4*>(W2t)[(tile * 3 + J) * 64 + lane];
  __shared__ __attribute__((aligned(16))) float4 wl[(7 - kTR > 0 ? 7 - kTR : 1) * 3 * 64];
  for (int e = tid; e < (7 - kTR) * 3 * 64; e += blockDim.x) wl[e] = reinterpret_cast<const float4*>(W2t)[kTR * 3 * 64 + e];
  // BN2's scale / shift: 4 distinct float4 per 16-channel group and wave -> LDS (24 registers less; 6 broadcast reads per tile)
  __shared__ __attribute__((aligned(16))) float bn_l[96];
  if (tid < 48) {
    bn_l[tid] = scale2[tid];
    bn_l[48 + tid] = shift2[tid];
  }
  __syncthreads();
  // complete on every path into the loops (see conv3x3_fwd_kernel: otherwise the first MFMAs of every row wait for the prefetch)
  __builtin_amdgcn_s_waitcnt(0x0F70);

  const int nbands = (H + band - 1) / band, nitems = B * nbands;
  const int xcol = 16 * TX * wave + x;                  // this lane's pixel column in tile 0
  // BatchNorm statistics: f64 running sums per (wave, channel) in LDS, updated once per stored row by the lane that owns the
  // channel after a DPP reduction over the 16 pixel lanes.  As 12 VGPRs of doubles they were spilled across the row's MFMA
  // stream, and the reload at the end of every row (s_waitcnt vmcnt, in order) waited for the next row's z prefetch as well.
  if (tid < kNWMax * 12 * 2) (&red[0][0][0])[tid] = 0.0;
  __syncthreads();

  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / nbands, bd = item - b * nbands;
    const int y0 = bd * band, y1 = min(y0 + band, H);
    const int ys = max(y0 - 1, 0);
    const float* zb = Z + ((size_t)b * H * W + xcol) * 48 + 4 * kk;      // + (row * W + 16 ti) * 48 + 16 J
    float* xb = X + ((size_t)b * H * W + xcol) * ldx + c_out0 + 3 * kk;  // + (row * W + 16 ti) * ldx
    float wa[TX][3], wb[TX][3], wc[TX][3];               // output rows yi - 1, yi, yi + 1
#pragma unroll
    for (int ti = 0; ti < TX; ++ti)
#pragma unroll
      for (int o = 0; o < 3; ++o) wa[ti][o] = wb[ti][o] = wc[ti][o] = 0.f;

    // z of (row, tile), requested one tile ahead (rows clamped into the image: unconditional loads)
    float4 zn[3] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(1.f, 2.f, 3.f, 4.f), make_float4(1.f, 2.f, 3.f, 4.f)};   // (TP_NOLOAD's operand)
    auto zload = [&](int row, int ti) {
#ifdef TP_NOLOAD   // experiment build: no z traffic (wrong results)
      return;
#endif
      const float* q = zb + ((size_t)min(row, H - 1) * W + 16 * ti) * 48;
#pragma unroll
      for (int J = 0; J < 3; ++J) zn[J] = *reinterpret_cast<const float4*>(q + 16 * J);
    };
    zload(ys, 0);

    for (int yi = ys; yi <= y1; ++yi) {
      const int par = yi & 1;
      if (yi < H) {
        const int mode = (yi < y0) ? kTop : (yi == y1) ? kBot : kFull;   // workgroup-uniform
        auto row = [&](auto mode_tag) {
          constexpr int MODE = decltype(mode_tag)::value;
          constexpr int T0 = MODE == kBot ? 4 : 0, T1 = MODE == kTop ? 3 : 7;
          // The 7 row tiles in two groups, A = tiles 0..3 (slots 0..15) and B = tiles 4..6 (slots 16..26), software-pipelined:
          // the shift-and-add of a group runs while the OTHER group's MFMAs are in flight (an accumulator is readable ~8
          // passes after its last MFMA: scattered right behind its own chain, every tile stalled on that).
          f32x4 d[7];
          // what pixel 0 of the first tile / pixel 15 of the last one owe the neighbour waves: EVERY lane stores (no exec-masked
          // block inside the MFMA stream), the edge lanes into the compact exchange buffer, the others into a scratch strip
          float* eLp = x == 0 ? &edge[par][wave][0][0][kk] : &junk[lane];
          float* eRp = x == 15 ? &edge[par][wave][1][0][kk] : &junk[lane];
          // shift-and-add of slots [s0, s1) of tile ti: slot s = 3 tap + o, tap = 3 (dy + 1) + (dx + 1); input row yi feeds
          // output row yi - dy, input pixel q output pixel q - dx
          auto scatter = [&](int ti, auto s0_tag, auto s1_tag) {
            constexpr int S0 = decltype(s0_tag)::value, S1 = decltype(s1_tag)::value;
#ifdef TP_NOSCATTER   // experiment build: the accumulators are summed without shifts (wrong results)
#pragma unroll
            for (int s = S0; s < S1; ++s) wb[ti][s % 3] += d[s / 4][s % 4];
            return;
#endif
#pragma unroll
            for (int s = S0; s < S1; ++s) {
              const int tap = s / 3, o = s % 3, dy = tap / 3 - 1, dx = tap % 3 - 1;
              const float v = d[s / 4][s % 4];
              auto add = [&](int tj, float val) {   // (dy, tj, o are constants after unrolling)
                if (dy < 0) wc[tj][o] += val;
                else if (dy == 0) wb[tj][o] += val;
                else wa[tj][o] += val;
              };
              if (dx == 0) {
                add(ti, v);
              } else if (dx < 0) {
                add(ti, shr1(v));
                if (ti + 1 < TX) add(ti + 1, shl15(v));
                else eRp[4 * (3 * (dy + 1) + o)] = v;
              } else {
                add(ti, shl1(v));
                if (ti > 0) add(ti - 1, shr15(v));
                else eLp[4 * (3 * (dy + 1) + o)] = v;
              }
            }
          };
          constexpr int A0 = T0 < 4 ? T0 : 4, A1 = T1 < 4 ? T1 : 4;   // group A's tiles of this mode
          constexpr int B0 = T0 > 4 ? T0 : 4, B1 = T1 > 4 ? T1 : 4;   // group B's
          using SA0 = std::integral_constant<int, 4 * A0>;
          using SA1 = std::integral_constant<int, 4 * A1>;
          using SB0 = std::integral_constant<int, 4 * B0>;
          using SB1 = std::integral_constant<int, (4 * B1 < 27 ? 4 * B1 : 27)>;
          float4 a[3];
          auto mfmas = [&](auto t0_tag, auto t1_tag) {
            constexpr int TA = decltype(t0_tag)::value, TB = decltype(t1_tag)::value;
#pragma unroll
            for (int tile = TA; tile < TB; ++tile) d[tile] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int J = 0; J < 3; ++J) {
              float4 wq[7];   // this 16-channel group's fragments of the LDS-resident row tiles
#pragma unroll
              for (int tile = (TA > kTR ? TA : kTR); tile < TB; ++tile) wq[tile] = wl[((tile - kTR) * 3 + J) * 64 + lane];
#pragma unroll
              for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int tile = TA; tile < TB; ++tile) {
                  float wv;
                  if (tile < kTR) wv = f4c(wf[tile < kTR ? tile : 0][J], t);
                  else wv = f4c(wq[tile], t);
                  d[tile] = mfma16(wv, f4c(a[J], t), d[tile]);
                }
            }
          };
#pragma unroll
          for (int ti = 0; ti < TX; ++ti) {
#pragma unroll
            for (int J = 0; J < 3; ++J) {
              const float4 sc = *reinterpret_cast<const float4*>(bn_l + 16 * J + 4 * kk);
              const float4 sh = *reinterpret_cast<const float4*>(bn_l + 48 + 16 * J + 4 * kk);
              a[J].x = fmaf(zn[J].x, sc.x, sh.x);
              a[J].y = fmaf(zn[J].y, sc.y, sh.y);
              a[J].z = fmaf(zn[J].z, sc.z, sh.z);
              a[J].w = fmaf(zn[J].w, sc.w, sh.w);
            }
            if (ti + 1 < TX) zload(yi, ti + 1);
            else zload(yi + 1, 0);
            __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of the MFMAs
            if constexpr (A1 > A0) mfmas(std::integral_constant<int, A0>{}, std::integral_constant<int, A1>{});
            if constexpr (B1 > B0)
              if (ti > 0) scatter(ti - 1, SB0{}, SB1{});     // under group A's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (B1 > B0) mfmas(std::integral_constant<int, B0>{}, std::integral_constant<int, B1>{});
            if constexpr (A1 > A0) scatter(ti, SA0{}, SA1{});   // under group B's MFMAs
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (B1 > B0) scatter(TX - 1, SB0{}, SB1{});
        };
        if (mode == kFull) row(std::integral_constant<int, kFull>{});
        else if (mode == kTop) row(std::integral_constant<int, kTop>{});
        else row(std::integral_constant<int, kBot>{});
#ifndef TP_NOEDGE   // experiment build: no barrier, no exchange (wrong at the waves' seams): what they cost
        if (NW > 1) {
          eml::lds_barrier();
          // the left neighbour's right-edge pixel feeds this wave's pixel 0, the right neighbour's left edge pixel 15 of
          // the last tile (values of the other lanes: 0)
          const bool lv = x == 0 && wave > 0, rv = x == 15 && wave + 1 < NW;
          const float* el = &edge[par][max(wave - 1, 0)][1][0][kk];   // the left neighbour's pixel 15, this lane group
          const float* er = &edge[par][min(wave + 1, NW - 1)][0][0][kk];   // the right neighbour's pixel 0
          // unconditional reads (every lane's address is valid), then a select: left to itself the compiler turns the
          // conditional expression into 18 exec-masked loads in 18 basic blocks
          auto pick = [&](const float* q, bool take) {
            float t = *q;
            asm volatile("" : "+v"(t));
            return take ? t : 0.f;
          };
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            wc[0][o] += pick(el + 4 * (0 + o), lv);
            wb[0][o] += pick(el + 4 * (3 + o), lv);
            wa[0][o] += pick(el + 4 * (6 + o), lv);
            wc[TX - 1][o] += pick(er + 4 * (0 + o), rv);
            wb[TX - 1][o] += pick(er + 4 * (3 + o), rv);
            wa[TX - 1][o] += pick(er + 4 * (6 + o), rv);
          }
        }
#endif
      }
      // output row yi - 1 is complete
      const int yo = yi - 1;
      if (yo >= y0 && yo < y1) {
        float ls[3] = {0.f, 0.f, 0.f}, lq[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < TX; ++ti) {
          float* dst = xb + ((size_t)yo * W + 16 * ti) * ldx;
#ifndef TP_NOSTORE   // experiment build: nothing written
          struct __attribute__((packed, aligned(4))) f3 { float a, b, c; };   // one global_store_dwordx3 (4-byte aligned)
          *reinterpret_cast<f3*>(dst) = f3{wa[ti][0], wa[ti][1], wa[ti][2]};
#endif
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            const float v = wa[ti][o];
            ls[o] += v;
            lq[o] = fmaf(v, v, lq[o]);
          }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float t1 = eml::row16_sum(ls[o]), t2 = eml::row16_sum(lq[o]);   // over the wave's 16 TX pixels of the row
          if (x == 0) {   // wave-private slots: no other lane touches them
            red[wave][3 * kk + o][0] += (double)t1;
            red[wave][3 * kk + o][1] += (double)t2;
          }
        }
      }
#pragma unroll
      for (int ti = 0; ti < TX; ++ti)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          wa[ti][o] = wb[ti][o];
          wb[ti][o] = wc[ti][o];
          wc[ti][o] = 0.f;
        }
    }
  }
  // channel statistics of the 12 new channels: the waves' running sums
  __syncthreads();
  if (tid < 32) {
    double t = 0.0;
    if (tid < 24)
      for (int w8 = 0; w8 < NW; ++w8) t += red[w8][tid >> 1][tid & 1];
    partials[(size_t)blockIdx.x * 32 + tid] = t;
  }
}

__global__ void permute_w2_tp_kernel(const float* __restrict__ W2, float* __restrict__ W2t) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 7 * 3 * 64 * 4; e += gridDim.x * blockDim.x) W2t[e] = eml::w2t_value(W2, e);
}

// TX (tiles per wave) and NW (waves side by side) for an image width, or 0
__host__ inline int tp_tiles(int W, int* nw) {
  if (W < 16 || (W & 15)) return 0;
  const int n = W / 16;
  for (int tx = 5; tx >= 4; --tx)
    if (n % tx == 0) {
      const int w = n / tx;
      if (w == 1 || w == 2 || w == 4) {   // 64 * NW <= 256 threads (__launch_bounds__)
        *nw = w;
        return tx;
      }
    }
  return 0;
}

}  // namespace

extern "C" int eml_dense_permute_w2_tp_f32(const float* W2, float* W2t, eml_stream_t stream) {
  if (!W2 || !W2t) return eml::fail(EML_EINVAL, "eml_dense_permute_w2_tp_f32: bad arguments");
  hipLaunchKernelGGL(permute_w2_tp_kernel, dim3(21), dim3(256), 0, (hipStream_t)stream, W2, W2t);
  return eml::check_launch("eml_dense_permute_w2_tp_f32");
}

extern "C" int eml_dense_conv3x3_fwd_tp_supported(int B, int H, int W) {
  int nw = 0;
  return (B >= 1 && H >= 1 && tp_tiles(W, &nw) != 0) ? nw : 0;
}

extern "C" int eml_dense_conv3x3_fwd_tp_f32(const float* Z, const float* scale2, const float* shift2, const float* W2t,
                                            float* X, int ldx, int c_out0, int B, int H, int W, int band_rows,
                                            double* partials, int grid, eml_stream_t stream) {
  if (!Z || !scale2 || !shift2 || !W2t || !X || !partials || B < 1 || H < 1 || W < 1 || grid < 1 || band_rows < 1 ||
      c_out0 + 12 > ldx || c_out0 < 0)
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_fwd_tp_f32: bad arguments");
  int nw = 0;
  const int tx = tp_tiles(W, &nw);
  if (!tx)
    return eml::fail(EML_EINVAL, "eml_dense_conv3x3_fwd_tp_f32: W = %d is not 16 * {4,5} * {1,2,4} (eml_dense_conv3x3_fwd_tp_supported)", W);
  // every one of the `grid` workgroups writes its partial row (zeros when it owns no band): bn_prepare folds `grid` rows
  if (tx == 5)
    hipLaunchKernelGGL(conv3x3_fwd_tp_kernel<5>, dim3(grid), dim3(64 * nw), 0, (hipStream_t)stream, Z, scale2, shift2, W2t, X, ldx,
                       c_out0, B, H, W, band_rows, partials);
  else
    hipLaunchKernelGGL(conv3x3_fwd_tp_kernel<4>, dim3(grid), dim3(64 * nw), 0, (hipStream_t)stream, Z, scale2, shift2, W2t, X, ldx,
                       c_out0, B, H, W, band_rows, partials);
  return eml::check_launch("eml_dense_conv3x3_fwd_tp_f32");
}
