// SphereConv2D for the few-channel OUTPUT layers, second generation (round 6): project, then gather.
//
// csrc/sphere_conv_narrow.hip gathers the 9 taps x 4 bilinear corners of every output pixel at full channel width and then
// multiplies by W2: 36 16-byte loads per 64 channels and pixel -- 9.2 KB of L1 traffic per pixel of conv_img (64 -> 3 at
// 128 x 256), which is what bounds it (0.80 ms forward, 0.94 ms weight gradient at B = 32: 0.3 TB/s of HBM traffic; tools/
// bench_narrow.py).  With O <= 4 output channels the product is cheaper the other way round -- the convolution is linear and
// the bilinear weights do not depend on the channel (reference: sphere_cnn.py:111-124, grid_sample then conv2d(stride 3)):
//
//   forward :  P[q][tap][o] = sum_c W2[o][tap*C + c] * X[q][c]      for every SOURCE pixel q   (an M x C x 36 GEMM, f32 MFMA)
//              Y[m][o]      = bias[o] + sum_tap sum_e wgt[p,tap,e] * P[idx[p,tap,e]][tap][o]      (36 16-byte gathers per PIXEL)
//   wgrad   :  V[q][tap][o] = sum_s twgt[q,tap,s] * dY[tidx[q,tap,s]][o]   (the transposed tap table: what every output pixel
//                                                                           sends back to source pixel q through tap `tap`)
//              dW2[o][tap*C + c] = sum_q V[q][tap][o] * X[q][c]             (a 36 x C GEMM with K = pixels, f32 MFMA, split-K)
//
// P / V are (pixels, 36) scratch tensors, column n = 4 * tap + o (o < 4; columns o >= O are zero): 144 bytes per source pixel
// against the 256 ... 2048 bytes of its X row.  Every sum runs in a fixed order (no atomics); the summation ORDER differs from the
// first generation's (channels first, corners second), results agree to f32 round-off.
#include <algorithm>

#include "eml_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kMaxC = 512;
constexpr int kCols = 36;   // 9 taps x 4 (padded) output channels

__host__ __device__ inline bool narrow2_supported(int C, int O) { return C >= 64 && C % 64 == 0 && C <= kMaxC && O >= 1 && O <= 4; }

// ---------------------------------------------------------------------------------------------------------------- project
// D^T form: MFMA rows = the 36 (48) columns n of P, MFMA columns = 16 source pixels, k = channels.  Lane (r, kk) loads the
// float4 of channels 16 j + 4 kk .. + 3 of pixel r (k is a summation label: step t of group j uses channel 16 j + 4 kk + t on
// both operands) and holds D rows 4 kk .. 4 kk + 3 of pixel r: 16-byte stores.  The weights sit in LDS in fragment order,
// Wl[((j * 3 + nt) * 4 + t) * 64 + lane] = W2[o][tap * C + 16 j + 4 (lane >> 4) + t]  with  n = 16 nt + (lane & 15) = 4 tap + o.
// A wave's next group of 64 channels (or its next tile's first) is requested before the current one's MFMAs.
__global__ __launch_bounds__(256, 2) void narrow_project_kernel(const float* __restrict__ X, const float* __restrict__ W2,
                                                                float* __restrict__ P, int Mq, int C, int O) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, kk = lane >> 4;
  const int nj = C >> 4;
  for (int e = tid; e < nj * 3 * 4 * 64; e += 256) {
    const int l = e & 63, t = (e >> 6) & 3, rest = e >> 8, nt = rest % 3, j = rest / 3;
    const int n = 16 * nt + (l & 15), c = 16 * j + 4 * (l >> 4) + t;
    const int tap = n >> 2, o = n & 3;
    Wl[e] = (n < kCols && o < O) ? W2[(size_t)o * 9 * C + (size_t)tap * C + c] : 0.f;
  }
  __syncthreads();
  const int wave = blockIdx.x * 4 + (tid >> 6), nwaves = gridDim.x * 4;
  const int ntiles = (Mq + 15) >> 4;
  const int nch = C >> 6;                       // chunks of 64 channels (4 groups j) per tile
  const int mine = wave < ntiles ? (ntiles - wave + nwaves - 1) / nwaves : 0;
  const int total = mine * nch;
  if (total == 0) return;
  auto fetch = [&](int it, f32x4 (&xv)[4]) {   // chunk `it` of this wave's (tile, chunk) sequence, clamped
    const int itc = min(it, total - 1);
    const int ti = itc / nch, ch = itc - ti * nch;
    const int q = min((wave + ti * nwaves) * 16 + r, Mq - 1);
    const float* xp = X + (size_t)q * C + 64 * ch + 4 * kk;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) xv[jj] = *reinterpret_cast<const f32x4*>(xp + 16 * jj);
  };
  f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  auto compute = [&](int it, const f32x4 (&xv)[4]) {
    const int ti = it / nch, ch = it - ti * nch;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float* wl = Wl + (size_t)(((4 * ch + jj) * 3) * 4) * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[nt] = mfma16(wl[(nt * 4 + t) * 64], xv[jj][t], acc[nt]);
    }
    if (ch == nch - 1) {
      const int q = (wave + ti * nwaves) * 16 + r;
      if (q < Mq) {
        float* dst = P + (size_t)q * kCols + 4 * kk;
        *reinterpret_cast<f32x4*>(dst) = acc[0];
        *reinterpret_cast<f32x4*>(dst + 16) = acc[1];
        if (kk == 0) *reinterpret_cast<f32x4*>(dst + 32) = acc[2];   // columns 32 .. 35
      }
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 xa[4], xb[4];
  fetch(0, xa);
  for (int it = 0; it < total; it += 2) {
    fetch(it + 1, xb);
    __builtin_amdgcn_sched_barrier(0);
    compute(it, xa);
    __builtin_amdgcn_sched_barrier(0);
    fetch(it + 2, xa);
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < total) compute(it + 1, xb);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------------- forward: gather
// One thread per output pixel: 9 table entries, 36 gathers of the float4 P[q][tap][0..3], O floats out.
// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: XCD k takes a CONTIGUOUS band of 256-pixel blocks
// (gridDim.x = 8 * blocks per XCD), so the rows of P its pixels gather from -- shared by up to 36 neighbours -- stay in one L2
// instead of being fetched into all eight.
template <int O>
__global__ __launch_bounds__(256) void narrow_gather_fwd_kernel(const float* __restrict__ P, const int* __restrict__ idx,
                                                                const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                float* __restrict__ Y, int M, int HW, int Po) {
  const int blk = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  {
    const int m = blk * 256 + threadIdx.x;
    if (m >= M) return;
    const int b = m / Po, p = m - b * Po;
    const float* pb = P + (size_t)b * HW * kCols;
    i32x4 id[9];
    f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      id[t] = *reinterpret_cast<const i32x4*>(idx + ((size_t)p * 9 + t) * 4);
      w[t] = *reinterpret_cast<const f32x4*>(wgt + ((size_t)p * 9 + t) * 4);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      f32x4 v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)   // corners off the map: index -1 and weight 0 in the table; the clamped address is finite
        v[e] = *reinterpret_cast<const f32x4*>(pb + (size_t)max(id[t][e], 0) * kCols + 4 * t);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float we = id[t][e] >= 0 ? w[t][e] : 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = fmaf(we, v[e][o], acc[o]);
      }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) Y[(size_t)m * O + o] = acc[o] + (bias ? bias[o] : 0.f);
  }
}

// ------------------------------------------------------------------------------------- weight gradient: V, then V^T X
// One thread per (source pixel, tap): the entries of the transposed tap table, ke per (pixel, tap), -1 = empty slot, four at a
// time (16-byte loads; slots 4..7 exist for the rows next to the poles only: rowmax says which pixels have any).  XCD bands
// as in the forward gather.
template <int O>
__global__ __launch_bounds__(256) void narrow_vgather_kernel(const float* __restrict__ dY, const int* __restrict__ tidx,
                                                             const float* __restrict__ twgt, int ke,
                                                             const unsigned char* __restrict__ rowmax, float* __restrict__ V,
                                                             long n /* B * HW * 9 */, int HW, int Po) {
  const long blk = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long i = blk * 256 + threadIdx.x;
  if (i >= n) return;
  const long qg = i / 9;
  const int t = (int)(i - qg * 9);
  const int b = (int)(qg / HW), q = (int)(qg - (long)b * HW);
  const int* ti = tidx + ((size_t)q * 9 + t) * ke;
  const float* tw = twgt + ((size_t)q * 9 + t) * ke;
  const float* gb = dY + (size_t)b * Po * O;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if ((ke & 3) == 0) {
    const int kq = (rowmax && rowmax[q] <= 4) ? 4 : ke;
    for (int s0 = 0; s0 < kq; s0 += 4) {
      const i32x4 ids = *reinterpret_cast<const i32x4*>(ti + s0);
      const f32x4 ws = *reinterpret_cast<const f32x4*>(tw + s0);
      float g[4][O];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < O; ++o) g[e][o] = gb[(size_t)max(ids[e], 0) * O + o];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = ids[e] >= 0 ? ws[e] : 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) v[o] = fmaf(w, g[e][o], v[o]);
      }
    }
  } else {
    for (int s = 0; s < ke; ++s) {
      const int pi = ti[s];
      const float w = pi >= 0 ? tw[s] : 0.f;
#pragma unroll
      for (int o = 0; o < O; ++o) v[o] = fmaf(w, gb[(size_t)max(pi, 0) * O + o], v[o]);
    }
  }
  *reinterpret_cast<f32x4*>(V + (size_t)qg * kCols + 4 * t) = v;
}

// partial[blockIdx.x][C][48]: G[c][n] = sum_q X[q][c] * V[q][n] over the pixels of this workgroup; blockIdx.y = the
// 64-channel group.  MFMA rows = channels (row r of tile t <-> channel 64 y + 4 r + t: the lane's float4 of X feeds four
// tiles), MFMA columns = n (three tiles), k = 4 pixels per step (lane group g = pixel).  The operands of a step are
// requested a step ahead of its MFMAs (two buffers), U k-steps per step.
__global__ __launch_bounds__(256, 2) void narrow_wgrad2_kernel(const float* __restrict__ X, const float* __restrict__ V,
                                                               float* __restrict__ partial, int Mq, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3][64 * 48] wave partials of waves 1..3
  constexpr int U = 4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
  const int c0 = 64 * blockIdx.y + 4 * r;
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct Operands {
    f32x4 x[U];
    float v[U][3];
    int okb;
  };
  const int nsteps = (Mq + 4 * U - 1) / (4 * U);
  const int n2 = 32 + min(r, 3);   // tile 2 holds columns 32 .. 35 (lanes r >= 4: a clamped address, masked below)
  auto issue = [&](int st, Operands& o) {
    o.okb = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = (st * U + u) * 4 + g;
      const bool ok = st < nsteps && q < Mq;
      const size_t qc = ok ? q : 0;
      o.okb |= ok ? (1 << u) : 0;
      o.x[u] = *reinterpret_cast<const f32x4*>(X + qc * C + c0);
      const float* vp = V + qc * kCols;
      o.v[u][0] = vp[r];
      o.v[u][1] = vp[16 + r];
      o.v[u][2] = vp[n2];
    }
  };
  auto run = [&](const Operands& o) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = (o.okb >> u) & 1;
      const f32x4 x = ok ? o.x[u] : f32x4{0.f, 0.f, 0.f, 0.f};
      const float b0 = o.v[u][0], b1 = o.v[u][1], b2 = r < 4 ? o.v[u][2] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] = mfma16(x[t], b0, acc[t][0]);
        acc[t][1] = mfma16(x[t], b1, acc[t][1]);
        acc[t][2] = mfma16(x[t], b2, acc[t][2]);
      }
    }
  };
  Operands oa, ob;
  issue(wave, oa);
  for (int st = wave; st < nsteps; st += 2 * nwaves) {
    issue(st + nwaves, ob);
    __builtin_amdgcn_sched_barrier(0);
    run(oa);
    __builtin_amdgcn_sched_barrier(0);
    issue(st + 2 * nwaves, oa);
    __builtin_amdgcn_sched_barrier(0);
    run(ob);
    __builtin_amdgcn_sched_barrier(0);
  }
  // D element i of tile (t, nt): row 4 g + i <-> channel 4 (4 g + i) + t of the group, column 16 nt + r
  if (wv > 0) {
    float* dst = smem + (size_t)(wv - 1) * 64 * 48;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[(4 * (4 * g + i) + t) * 48 + 16 * nt + r] = acc[t][nt][i];
  }
  __syncthreads();
  if (wv == 0) {
    float* out = partial + ((size_t)blockIdx.x * C + 64 * blockIdx.y) * 48;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = (4 * (4 * g + i) + t) * 48 + 16 * nt + r;
          out[e] = ((acc[t][nt][i] + smem[e]) + smem[64 * 48 + e]) + smem[2 * 64 * 48 + e];
        }
  }
}

// dW2[o][tap * C + c] = fixed-order sum over the S workgroup partials of G[c][4 tap + o]
__global__ __launch_bounds__(256) void narrow_wgrad2_reduce_kernel(const float* __restrict__ partial, int S, int C, int O,
                                                                   float* __restrict__ dW2) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;            // (c, n) with n < 48
  float s = 0.f;
  if (e < C * 48)
    for (int z = zl; z < S; z += 16) s += partial[(size_t)z * C * 48 + e];
  red[zl][el] = s;
  __syncthreads();
  if (zl == 0 && e < C * 48) {
    float tot = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) tot += red[z][el];
    const int c = e / 48, n = e - 48 * c, tap = n >> 2, o = n & 3;
    if (n < kCols && o < O) dW2[(size_t)o * 9 * C + (size_t)tap * C + c] = tot;
  }
}

// ------------------------------------------------------------------------------------------- input gradient: V, then V W
// dX[q][c] = sum_n V[q][n] * Wn[n][c],  Wn[4 tap + o][c] = W2[o][tap * C + c]: D^T form again -- MFMA rows = 16 channels,
// columns = 16 source pixels, k = the 36 columns of V: two groups of four steps fed from the lane's float4 V[q][16 j + 4 kk ..]
// (step t of group j uses column 16 j + 4 kk + t on both operands) and a ninth step for columns 32 .. 35.  Weights in LDS in
// fragment order; a lane holds 4 consecutive channels of its pixel: 16-byte stores.
__global__ __launch_bounds__(256, 2) void narrow_expand_kernel(const float* __restrict__ V, const float* __restrict__ W2,
                                                               float* __restrict__ dX, int Mq, int C, int O) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];   // [(2 * CT + ...)]: main [2][CT][4][64], tail [CT][64]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, kk = lane >> 4;
  const int CT = C >> 4;
  float* Wt = Wl + 2 * CT * 4 * 64;
  auto wn = [&](int n, int c) {
    const int tap = n >> 2, o = n & 3;
    return (n < kCols && o < O) ? W2[(size_t)o * 9 * C + (size_t)tap * C + c] : 0.f;
  };
  for (int e = tid; e < 2 * CT * 4 * 64; e += 256) {
    const int l = e & 63, t = (e >> 6) & 3, rest = e >> 8, ct = rest % CT, j = rest / CT;
    Wl[e] = wn(16 * j + 4 * (l >> 4) + t, 16 * ct + (l & 15));
  }
  for (int e = tid; e < CT * 64; e += 256) {
    const int l = e & 63, ct = e >> 6;
    Wt[e] = wn(32 + (l >> 4), 16 * ct + (l & 15));
  }
  __syncthreads();
  const int wave = blockIdx.x * 4 + (tid >> 6), nwaves = gridDim.x * 4;
  const int ntiles = (Mq + 15) >> 4;
  struct Tile {
    f32x4 v0, v1;
    float vt;
  };
  auto fetch = [&](int tile, Tile& t) {
    const int q = min(min(tile, ntiles - 1) * 16 + r, Mq - 1);
    const float* vp = V + (size_t)q * kCols;
    t.v0 = *reinterpret_cast<const f32x4*>(vp + 4 * kk);
    t.v1 = *reinterpret_cast<const f32x4*>(vp + 16 + 4 * kk);
    t.vt = vp[32 + kk];
  };
  auto compute = [&](int tile, const Tile& t) {
    const int q = tile * 16 + r;
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* w0 = Wl + (size_t)(ct * 4) * 64 + lane;
      const float* w1 = Wl + (size_t)((CT + ct) * 4) * 64 + lane;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc = mfma16(w0[tt * 64], t.v0[tt], acc);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc = mfma16(w1[tt * 64], t.v1[tt], acc);
      acc = mfma16(Wt[ct * 64 + lane], t.vt, acc);
      if (q < Mq) *reinterpret_cast<f32x4*>(dX + (size_t)q * C + 16 * ct + 4 * kk) = acc;
    }
  };
  Tile ta, tb;
  fetch(wave, ta);
  for (int tile = wave; tile < ntiles; tile += 2 * nwaves) {
    fetch(tile + nwaves, tb);
    __builtin_amdgcn_sched_barrier(0);
    compute(tile, ta);
    __builtin_amdgcn_sched_barrier(0);
    fetch(tile + 2 * nwaves, ta);
    __builtin_amdgcn_sched_barrier(0);
    if (tile + nwaves < ntiles) compute(tile + nwaves, tb);
    __builtin_amdgcn_sched_barrier(0);
  }
}

int narrow2_wgrad_blocks(long Mq) { return (int)std::max<long>(1, std::min<long>(512, (Mq + 255) / 256)); }

int launch_vgather(const char* what, const float* dY, const int* tidx, const float* twgt, int ke, const unsigned char* rowmax,
                   float* V, long Mq, int HW, int Po, int O, eml_stream_t stream);

#define EML_NARROW2_DISPATCH(O_, STMT)             \
  switch (O_) {                                    \
    case 1: { constexpr int OV = 1; STMT; } break; \
    case 2: { constexpr int OV = 2; STMT; } break; \
    case 3: { constexpr int OV = 3; STMT; } break; \
    default: { constexpr int OV = 4; STMT; } break; \
  }

int launch_vgather(const char* what, const float* dY, const int* tidx, const float* twgt, int ke, const unsigned char* rowmax,
                   float* V, long Mq, int HW, int Po, int O, eml_stream_t stream) {
  const long nv = Mq * 9;
  const long gridv = 8 * (((nv + 255) / 256 + 7) / 8);
  if (gridv > 2147483647L) return eml::fail(EML_EINVAL, "%s: too many pixels", what);
  EML_NARROW2_DISPATCH(O, {
    hipLaunchKernelGGL((narrow_vgather_kernel<OV>), dim3((unsigned)gridv), dim3(256), 0, (hipStream_t)stream, dY, tidx, twgt, ke,
                       rowmax, V, nv, HW, Po);
  })
  return eml::check_launch(what);
}

}  // namespace

extern "C" size_t eml_sphere_conv_narrow_scratch_floats(int B, int HW) {
  return (B < 1 || HW < 1) ? 0 : (size_t)B * HW * kCols;
}

extern "C" int eml_sphere_conv_narrow_fwd2_f32(const float* X, const int* idx, const float* wgt, const float* W2, const float* bias,
                                               float* Y, float* scratch, int B, int HW, int Po, int C, int O, eml_stream_t stream) {
  if (!X || !idx || !wgt || !W2 || !Y || !scratch || B < 0 || HW < 1 || Po < 1)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_fwd2_f32: null pointer or empty shape");
  if (!narrow2_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_fwd2_f32: need C %% 64 == 0, C <= %d, 1 <= O <= 4 (C=%d, O=%d)", kMaxC, C, O);
  const long M = (long)B * Po, Mq = (long)B * HW;
  if (M > 2147483647L || Mq * kCols > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_fwd2_f32: too many pixels");
  if (B == 0) return EML_OK;
  const size_t lds = (size_t)C * 48 * sizeof(float);
  const int gridp = (int)std::min<long>(512, (Mq + 63) / 64);
  EML_ENSURE_LDS((&narrow_project_kernel), lds);
  hipLaunchKernelGGL(narrow_project_kernel, dim3(gridp), dim3(256), lds, (hipStream_t)stream, X, W2, scratch, (int)Mq, C, O);
  int rc = eml::check_launch("eml_sphere_conv_narrow_fwd2_f32(project)");
  if (rc) return rc;
  const int gridg = (int)(8 * (((M + 255) / 256 + 7) / 8));   // whole bands of 256-pixel blocks per XCD
  EML_NARROW2_DISPATCH(O, {
    hipLaunchKernelGGL((narrow_gather_fwd_kernel<OV>), dim3(gridg), dim3(256), 0, (hipStream_t)stream, scratch, idx, wgt, bias, Y,
                       (int)M, HW, Po);
  })
  return eml::check_launch("eml_sphere_conv_narrow_fwd2_f32(gather)");
}

extern "C" size_t eml_sphere_conv_narrow_wgrad2_partial_floats(int B, int HW, int C) {
  if (B < 1 || HW < 1 || C < 64 || C % 64 || C > kMaxC) return 0;
  return (size_t)narrow2_wgrad_blocks((long)B * HW) * C * 48;
}

extern "C" int eml_sphere_conv_narrow_wgrad2_f32(const float* X, const int* tidx, const float* twgt, int ke,
                                                 const unsigned char* rowmax, const float* dY, float* scratch, float* partial, float* dW2, int B, int HW, int Po, int C, int O,
                                                 eml_stream_t stream) {
  if (!X || !tidx || !twgt || !dY || !scratch || !partial || !dW2 || B < 1 || HW < 1 || Po < 1 || ke < 1 || ke > 8)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_wgrad2_f32: null pointer, empty shape or ke outside 1..8");
  if (!narrow2_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_wgrad2_f32: need C %% 64 == 0, C <= %d, 1 <= O <= 4 (C=%d, O=%d)", kMaxC, C, O);
  const long Mq = (long)B * HW;
  if (Mq * kCols > 2147483647L || (long)B * Po > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_wgrad2_f32: too many pixels");
  int rc = launch_vgather("eml_sphere_conv_narrow_wgrad2_f32(V)", dY, tidx, twgt, ke, rowmax, scratch, Mq, HW, Po, O, stream);
  if (rc) return rc;
  const int blocks = narrow2_wgrad_blocks(Mq);
  const size_t lds = (size_t)3 * 64 * 48 * sizeof(float);
  hipLaunchKernelGGL(narrow_wgrad2_kernel, dim3(blocks, C / 64), dim3(256), lds, (hipStream_t)stream, X, scratch, partial, (int)Mq, C);
  rc = eml::check_launch("eml_sphere_conv_narrow_wgrad2_f32(product)");
  if (rc) return rc;
  hipLaunchKernelGGL(narrow_wgrad2_reduce_kernel, dim3((C * 48 + 15) / 16), dim3(256), 0, (hipStream_t)stream, partial, blocks, C, O,
                     dW2);
  return eml::check_launch("eml_sphere_conv_narrow_wgrad2_f32(reduce)");
}

// dX (B*HW, C) = V W: `scratch` holds V when scratch_has_v != 0 (eml_sphere_conv_narrow_wgrad2_f32 of the SAME dY left it there),
// else it is computed first.
extern "C" int eml_sphere_conv_narrow_dgrad2_f32(const float* dY, const int* tidx, const float* twgt, int ke,
                                                 const unsigned char* rowmax, const float* W2, float* dX, float* scratch,
                                                 int scratch_has_v, int B, int HW, int Po, int C, int O, eml_stream_t stream) {
  if (!dY || !tidx || !twgt || !W2 || !dX || !scratch || B < 0 || HW < 1 || Po < 1 || ke < 1 || ke > 8)
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_dgrad2_f32: null pointer, empty shape or ke outside 1..8");
  if (!narrow2_supported(C, O))
    return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_dgrad2_f32: need C %% 64 == 0, C <= %d, 1 <= O <= 4 (C=%d, O=%d)", kMaxC, C, O);
  const long Mq = (long)B * HW;
  if (Mq * kCols > 2147483647L || (long)B * Po > 2147483647L) return eml::fail(EML_EINVAL, "eml_sphere_conv_narrow_dgrad2_f32: too many pixels");
  if (Mq == 0) return EML_OK;
  if (!scratch_has_v) {
    const int rc = launch_vgather("eml_sphere_conv_narrow_dgrad2_f32(V)", dY, tidx, twgt, ke, rowmax, scratch, Mq, HW, Po, O, stream);
    if (rc) return rc;
  }
  const size_t lds = (size_t)C * kCols * sizeof(float);
  const int grid = (int)std::min<long>(512, (Mq + 63) / 64);
  EML_ENSURE_LDS((&narrow_expand_kernel), lds);
  hipLaunchKernelGGL(narrow_expand_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, scratch, W2, dX, (int)Mq, C, O);
  return eml::check_launch("eml_sphere_conv_narrow_dgrad2_f32(expand)");
}
