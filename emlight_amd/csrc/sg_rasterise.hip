// Spherical-Gaussian lobes -> equirectangular panorama, forward and d/d(colors).
//
// Replaces the per-light Python loop of convert_to_panorama
// (reference RegressionNetwork/util.py:222-245): there each of the N lights costs ~6 ATen
// launches and a full re-read/re-write of the (B,3,H,W) accumulator.  Here one launch keeps
// the three channel accumulators of a pixel in registers, streams the lights of the sample
// through LDS (broadcast reads), and writes the panorama exactly once.
//
// Roofline: B*N*H*W exp2 evaluations against B*3*H*W*4 output bytes.  At N=128 that is
// 128 transcendentals per 12 output bytes -> instruction-bound, not HBM-bound -- and most of
// them are exactly zero: a lobe of size .0025 underflows (exp2(t) == 0 for t < -150 in f32)
// outside a 42-degree cone.  Round 2 tested every (wave, light) pair for that inside the
// accumulation loop (two LDS reads, a dot product, a ballot and a loop-carried branch per
// pair: the loop was bound by the TEST, not by the exponentials).  Now the cull is
// hierarchical: each wave first builds the list of lights that can reach its 16x8-pixel patch
// at all -- one cone test per light against the patch's bounding cap (centre direction c,
// chord radius r: dot(L, p) <= dot(L, c) + |L| r for every pixel p of the patch), 64 lights
// per instruction, survivors compacted in index order into a wave-private LDS array -- and
// the accumulation loop then walks only the survivors, branch-free, two pixels per lane.
// A light that is not on the list contributes exp2(t) == 0 to every pixel of the patch, so
// the result is bit-identical to evaluating all N lights in order (tests: the exhaustive
// variant of this kernel, flag EML_SG_EXHAUSTIVE).
#include "eml_common.h"

namespace {

constexpr int kTileW = 32;   // block tile: 2x2 waves, each wave a 16(w) x 8(h) pixel patch, two rows per lane
constexpr int kTileH = 16;
constexpr int kChunk = 128;  // lights staged per LDS pass (8 floats each = 4 KiB, + 4 KiB of survivors per wave)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kCull = -150.0f;  // exp2(-150) underflows below the smallest f32 denormal

struct __attribute__((aligned(16))) Lobe {
  float dx, dy, dz, k2;  // direction, log2(e)/size
  float r, g, b, pad;    // colour
};

template <bool kExhaustive, bool kCount>
__global__ __launch_bounds__(256) void sg_rasterise_kernel(
    const float* __restrict__ dirs, const float* __restrict__ sizes,
    const float* __restrict__ colors, float* __restrict__ out, int N, int H, int W, float step,
    unsigned long long* __restrict__ executed) {
  __shared__ Lobe lobes[kChunk];
  __shared__ Lobe kept[4][kChunk];
  __shared__ float sin_t[kTileH], cos_t[kTileH], sin_p[kTileW], cos_p[kTileW];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  // f32 view vectors of the tile's pixels on the H x 2H grid, evaluated like the reference (util.py:223-233):
  // theta = (h+.5)*f32(pi/H), phi = (w+.5)*f32(pi/H), xyz = (sin th cos ph, sin th sin ph, cos th); one sincos per
  // tile row / column (48 per block) instead of three per lane
  // the first chunk of lights is REQUESTED before anything else: its global-memory latency hides behind the sincos below
  // (later chunks are requested while the previous one is being accumulated)
  Lobe pre;
  auto preload = [&](int base) {
    const int li_ = min(base + tid, N - 1);
    const size_t li = (size_t)b * N + li_;
    pre.dx = dirs[3 * li + 0];
    pre.dy = dirs[3 * li + 1];
    pre.dz = dirs[3 * li + 2];
    pre.k2 = sizes[li];          // divided when it is committed to LDS
    pre.r = colors[3 * li + 0];
    pre.g = colors[3 * li + 1];
    pre.b = colors[3 * li + 2];
    pre.pad = 0.f;
  };
  if (tid < kChunk) preload(0);
  if (tid < kTileH) sincosf(((float)(blockIdx.y * kTileH + tid) + 0.5f) * step, &sin_t[tid], &cos_t[tid]);
  else if (tid < kTileH + kTileW)
    sincosf(((float)(blockIdx.x * kTileW + tid - kTileH) + 0.5f) * step, &sin_p[tid - kTileH], &cos_p[tid - kTileH]);
  __syncthreads();
  const int cw = (wave & 1) * 16 + (lane & 15), r0 = (wave >> 1) * 8 + (lane >> 4), r1 = r0 + 4;
  const int w = blockIdx.x * kTileW + cw;
  const int h0 = blockIdx.y * kTileH + r0, h1 = h0 + 4;
  const float p0x = sin_t[r0] * cos_p[cw], p0y = sin_t[r0] * sin_p[cw], p0z = cos_t[r0];
  const float p1x = sin_t[r1] * cos_p[cw], p1y = sin_t[r1] * sin_p[cw], p1z = cos_t[r1];
  // bounding cap of the wave's 128 pixels: centre = midpoint of pixels (row 3, col 8) and (row 4, col 7), radius = the
  // largest chord to any of the pixels (a reduction over the lanes, no geometric argument needed)
  float cx, cy, cz, rr;
  {
    auto lane_of = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    cx = lane_of(p0x, 56) + lane_of(p1x, 7);
    cy = lane_of(p0y, 56) + lane_of(p1y, 7);
    cz = lane_of(p0z, 56) + lane_of(p1z, 7);
    const float inv = rsqrtf(fmaf(cz, cz, fmaf(cy, cy, cx * cx)));
    cx *= inv, cy *= inv, cz *= inv;
    const float a0 = p0x - cx, a1 = p0y - cy, a2 = p0z - cz, b0 = p1x - cx, b1 = p1y - cy, b2 = p1z - cz;
    const float d2 = fmaxf(fmaf(a2, a2, fmaf(a1, a1, a0 * a0)), fmaf(b2, b2, fmaf(b1, b1, b0 * b0)));
    rr = sqrtf(eml::wave_max_dpp(d2)) * 1.0001f + 1e-6f;
  }

  // the lane's two pixels travel as register pairs: every step of the lobe evaluation is ONE packed instruction
  // (v_pk_mul / v_pk_fma / v_pk_add_f32) for both -- component-wise IEEE fma, bit-identical to the scalar chain
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f px = v2f{p0x, p1x}, py = v2f{p0y, p1y}, pz = v2f{p0z, p1z};
  v2f ar = v2f{0.f, 0.f}, ag = ar, ab = ar;
  unsigned long long n_exec = 0;
  for (int base = 0; base < N; base += kChunk) {
    const int cnt = min(kChunk, N - base);
    if (base > 0) __syncthreads();   // the previous chunk's lobes / survivor lists are no longer read
    if (tid < cnt) {
      Lobe L = pre;
      L.k2 = kLog2e / pre.k2;
      lobes[tid] = L;
    }
    __syncthreads();
    if (base + kChunk < N && tid < kChunk) preload(base + kChunk);
    int n = cnt;
    const Lobe* src = lobes;
    if (!kExhaustive) {
      n = 0;
      for (int j = 0; j < cnt; j += 64) {
        const int li = j + lane;
        const Lobe L = lobes[min(li, cnt - 1)];
        const float len = sqrtf(fmaf(L.dz, L.dz, fmaf(L.dy, L.dy, L.dx * L.dx)));
        // upper bound of dot(L, p) over the patch, with slack far above the f32 rounding of either side
        const float s = fmaf(L.dz, cz, fmaf(L.dy, cy, L.dx * cx)) + fmaf(len, rr, 1e-5f * (1.0f + len));
        const bool odd = !(L.k2 > 0.f) || L.k2 > 3.0e38f || !(len < 3.0e38f);   // non-positive / zero sizes, NaNs: never cull
        const bool keep = li < cnt && (odd || !((s - 1.0f) * L.k2 <= kCull));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep) kept[wave][pos] = L;   // ascending light index: the accumulation order of the exhaustive loop
        n += __builtin_popcountll(m);
      }
      src = kept[wave];
      // the list is wave-private and a wave's LDS operations execute in order: only the compiler needs a fence
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (kCount) n_exec += (unsigned long long)n * 128ull;
#pragma unroll 2
    for (int i = 0; i < n; ++i) {
      const Lobe L = src[i];  // same address on every lane: LDS broadcast
      const v2f dx = v2f{L.dx, L.dx}, dy = v2f{L.dy, L.dy}, dz = v2f{L.dz, L.dz}, k2 = v2f{L.k2, L.k2};
      const v2f dot = __builtin_elementwise_fma(dz, pz, __builtin_elementwise_fma(dy, py, dx * px));
      const v2f t = (dot - v2f{1.0f, 1.0f}) * k2;
      const v2f e = v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
      ar = __builtin_elementwise_fma(v2f{L.r, L.r}, e, ar);
      ag = __builtin_elementwise_fma(v2f{L.g, L.g}, e, ag);
      ab = __builtin_elementwise_fma(v2f{L.b, L.b}, e, ab);
    }
  }
  if (kCount && lane == 0 && executed) atomicAdd(executed, n_exec);
  const size_t plane = (size_t)H * W;
  if (w < W) {
    if (h0 < H) {
      float* o = out + (size_t)b * 3 * plane + (size_t)h0 * W + w;
      o[0] = ar.x;
      o[plane] = ag.x;
      o[2 * plane] = ab.x;
    }
    if (h1 < H) {
      float* o = out + (size_t)b * 3 * plane + (size_t)h1 * W + w;
      o[0] = ar.y;
      o[plane] = ag.y;
      o[2 * plane] = ab.y;
    }
  }
}

// d/d(colors): one block per (sample, group of kLG lights); the block sweeps the whole
// panorama, each thread carrying kLG x 3 partial sums, then a wave-shuffle + LDS reduction.
constexpr int kLG = 8;

__global__ __launch_bounds__(256) void sg_rasterise_bwd_colors_kernel(
    const float* __restrict__ dirs, const float* __restrict__ sizes,
    const float* __restrict__ gout, float* __restrict__ gcolors, int N, int H, int W, float step) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sin_t = smem;          // [H]
  float* cos_t = sin_t + H;     // [H]
  float* sin_p = cos_t + H;     // [W]
  float* cos_p = sin_p + W;     // [W]
  float* red = cos_p + W;       // [4 waves][kLG*3]
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * kLG;
  const int tid = threadIdx.x;
  for (int i = tid; i < H; i += 256) sincosf(((float)i + 0.5f) * step, &sin_t[i], &cos_t[i]);
  for (int i = tid; i < W; i += 256) sincosf(((float)i + 0.5f) * step, &sin_p[i], &cos_p[i]);

  float dx[kLG], dy[kLG], dz[kLG], k2[kLG];
#pragma unroll
  for (int l = 0; l < kLG; ++l) {
    const int li = min(l0 + l, N - 1);
    const size_t gi = (size_t)b * N + li;
    dx[l] = dirs[3 * gi + 0];
    dy[l] = dirs[3 * gi + 1];
    dz[l] = dirs[3 * gi + 2];
    k2[l] = kLog2e / sizes[gi];
  }
  float acc[kLG][3];
#pragma unroll
  for (int l = 0; l < kLG; ++l) acc[l][0] = acc[l][1] = acc[l][2] = 0.f;
  __syncthreads();

  const size_t plane = (size_t)H * W;
  const float* g = gout + (size_t)b * 3 * plane;
  for (size_t p = tid; p < plane; p += 256) {
    const int h = (int)(p / W), w = (int)(p % W);
    const float px = sin_t[h] * cos_p[w], py = sin_t[h] * sin_p[w], pz = cos_t[h];
    const float g0 = g[p], g1 = g[plane + p], g2 = g[2 * plane + p];
#pragma unroll
    for (int l = 0; l < kLG; ++l) {
      const float dot = fmaf(dz[l], pz, fmaf(dy[l], py, dx[l] * px));
      const float e = __builtin_amdgcn_exp2f((dot - 1.0f) * k2[l]);
      acc[l][0] = fmaf(g0, e, acc[l][0]);
      acc[l][1] = fmaf(g1, e, acc[l][1]);
      acc[l][2] = fmaf(g2, e, acc[l][2]);
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int l = 0; l < kLG; ++l)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = eml::wave_sum(acc[l][c]);
      if (lane == 0) red[wave * kLG * 3 + l * 3 + c] = s;
    }
  __syncthreads();
  if (tid < kLG * 3) {
    const int l = tid / 3;
    if (l0 + l < N) {
      const float s = (red[tid] + red[kLG * 3 + tid]) + (red[2 * kLG * 3 + tid] + red[3 * kLG * 3 + tid]);
      gcolors[((size_t)b * N + l0 + l) * 3 + (tid % 3)] = s;
    }
  }
}

// d/d(colors) with the forward's hierarchical cull (round 4): same tiles, same bounding cap, same survivor lists.  A wave
// evaluates exp2 only for the lights that reach its 16 x 8 patch, reduces  sum_pixels gout * exp2(.)  over its lanes (three DPP
// wave sums per survivor) and parks the result in a wave-private LDS row indexed by the light (a light is on a wave's list at
// most once per chunk: a store, not an accumulation); the four waves are added per workgroup, and the per-tile partials
// [B][tiles][N][3] by a second kernel in tile order -- no atomics, run-to-run exact.  A culled light contributes exp2(t) == 0
// to every pixel of the patch, i.e. exactly 0: the result equals the exhaustive evaluation of the same sums.
__global__ __launch_bounds__(256) void sg_rasterise_bwd_colors_tiled_kernel(
    const float* __restrict__ dirs, const float* __restrict__ sizes, const float* __restrict__ gout,
    float* __restrict__ partial, int N, int H, int W, float step, int exhaustive) {
  __shared__ float4 lobes[kChunk];            // direction, log2(e)/size
  __shared__ float4 kept[4][kChunk];
  __shared__ int keptidx[4][kChunk];
  __shared__ float accw[4][kChunk][3];
  __shared__ float sin_t[kTileH], cos_t[kTileH], sin_p[kTileW], cos_p[kTileW];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  float4 pre = make_float4(0.f, 0.f, 0.f, 1.f);
  auto preload = [&](int base) {
    const size_t li = (size_t)b * N + min(base + tid, N - 1);
    pre = make_float4(dirs[3 * li + 0], dirs[3 * li + 1], dirs[3 * li + 2], sizes[li]);
  };
  if (tid < kChunk) preload(0);
  if (tid < kTileH) sincosf(((float)(blockIdx.y * kTileH + tid) + 0.5f) * step, &sin_t[tid], &cos_t[tid]);
  else if (tid < kTileH + kTileW)
    sincosf(((float)(blockIdx.x * kTileW + tid - kTileH) + 0.5f) * step, &sin_p[tid - kTileH], &cos_p[tid - kTileH]);
  __syncthreads();
  const int cw = (wave & 1) * 16 + (lane & 15), r0 = (wave >> 1) * 8 + (lane >> 4), r1 = r0 + 4;
  const int w = blockIdx.x * kTileW + cw;
  const int h0 = blockIdx.y * kTileH + r0, h1 = h0 + 4;
  const float p0x = sin_t[r0] * cos_p[cw], p0y = sin_t[r0] * sin_p[cw], p0z = cos_t[r0];
  const float p1x = sin_t[r1] * cos_p[cw], p1y = sin_t[r1] * sin_p[cw], p1z = cos_t[r1];
  float cx, cy, cz, rr;   // bounding cap of the wave's patch: exactly the forward's
  {
    auto lane_of = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    cx = lane_of(p0x, 56) + lane_of(p1x, 7);
    cy = lane_of(p0y, 56) + lane_of(p1y, 7);
    cz = lane_of(p0z, 56) + lane_of(p1z, 7);
    const float inv = rsqrtf(fmaf(cz, cz, fmaf(cy, cy, cx * cx)));
    cx *= inv, cy *= inv, cz *= inv;
    const float a0 = p0x - cx, a1 = p0y - cy, a2 = p0z - cz, b0 = p1x - cx, b1 = p1y - cy, b2 = p1z - cz;
    const float d2 = fmaxf(fmaf(a2, a2, fmaf(a1, a1, a0 * a0)), fmaf(b2, b2, fmaf(b1, b1, b0 * b0)));
    rr = sqrtf(eml::wave_max_dpp(d2)) * 1.0001f + 1e-6f;
  }
  // the lane's two pixels of the output gradient (0 off the map)
  const size_t plane = (size_t)H * W;
  const float* g = gout + (size_t)b * 3 * plane;
  float g0[3] = {0.f, 0.f, 0.f}, g1[3] = {0.f, 0.f, 0.f};
  if (w < W && h0 < H) {
    const size_t o = (size_t)h0 * W + w;
    g0[0] = g[o]; g0[1] = g[plane + o]; g0[2] = g[2 * plane + o];
  }
  if (w < W && h1 < H) {
    const size_t o = (size_t)h1 * W + w;
    g1[0] = g[o]; g1[1] = g[plane + o]; g1[2] = g[2 * plane + o];
  }
  const int tile = blockIdx.y * gridDim.x + blockIdx.x, ntiles = gridDim.x * gridDim.y;
  float* out = partial + ((size_t)b * ntiles + tile) * N * 3;
  for (int base = 0; base < N; base += kChunk) {
    const int cnt = min(kChunk, N - base);
    if (base > 0) __syncthreads();   // the previous chunk's lobes / lists / sums are no longer read
    if (tid < cnt) lobes[tid] = make_float4(pre.x, pre.y, pre.z, kLog2e / pre.w);
    for (int i = lane; i < kChunk * 3; i += 64) (&accw[wave][0][0])[i] = 0.f;
    __syncthreads();
    if (base + kChunk < N && tid < kChunk) preload(base + kChunk);
    int n = 0;
    for (int j = 0; j < cnt; j += 64) {
      const int li = j + lane;
      const float4 L = lobes[min(li, cnt - 1)];
      const float len = sqrtf(fmaf(L.z, L.z, fmaf(L.y, L.y, L.x * L.x)));
      const float s_ = fmaf(L.z, cz, fmaf(L.y, cy, L.x * cx)) + fmaf(len, rr, 1e-5f * (1.0f + len));
      const bool odd = !(L.w > 0.f) || L.w > 3.0e38f || !(len < 3.0e38f);   // as in the forward: never cull these
      const bool keep = li < cnt && (exhaustive || odd || !((s_ - 1.0f) * L.w <= kCull));
      const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
      const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      if (keep) {
        kept[wave][pos] = L;
        keptidx[wave][pos] = li;
      }
      n += __builtin_popcountll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = 0; i < n; ++i) {
      const float4 L = kept[wave][i];   // broadcast reads
      const int li = keptidx[wave][i];
      const float e0 = __builtin_amdgcn_exp2f((fmaf(L.z, p0z, fmaf(L.y, p0y, L.x * p0x)) - 1.0f) * L.w);
      const float e1 = __builtin_amdgcn_exp2f((fmaf(L.z, p1z, fmaf(L.y, p1y, L.x * p1x)) - 1.0f) * L.w);
      const float sr = eml::wave_sum_dpp(fmaf(g1[0], e1, g0[0] * e0));
      const float sg = eml::wave_sum_dpp(fmaf(g1[1], e1, g0[1] * e0));
      const float sb = eml::wave_sum_dpp(fmaf(g1[2], e1, g0[2] * e0));
      if (lane == 0) {
        accw[wave][li][0] = sr;
        accw[wave][li][1] = sg;
        accw[wave][li][2] = sb;
      }
    }
    __syncthreads();
    for (int i = tid; i < cnt * 3; i += 256) {
      const float* a0 = &accw[0][0][0];
      out[(size_t)base * 3 + i] = (a0[i] + a0[kChunk * 3 + i]) + (a0[2 * kChunk * 3 + i] + a0[3 * kChunk * 3 + i]);
    }
  }
}

__global__ __launch_bounds__(256) void sg_reduce_tiles_kernel(const float* __restrict__ partial, int ntiles, int n3,
                                                              float* __restrict__ gcolors, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;   // (b, light, channel)
  if (e >= total) return;
  const int b = e / n3, i = e - b * n3;
  const float* p = partial + (size_t)b * ntiles * n3 + i;
  float s = 0.f;
  for (int t = 0; t < ntiles; ++t) s += p[(size_t)t * n3];
  gcolors[e] = s;
}

}  // namespace

extern "C" int eml_sg_rasterise_ex_f32(const float* dirs, const float* sizes, const float* colors, float* out, int B,
                                       int N, int H, int W, int flags, unsigned long long* executed_exp,
                                       eml_stream_t stream) {
  if (!dirs || !sizes || !colors || !out) return eml::fail(EML_EINVAL, "eml_sg_rasterise_f32: null pointer");
  if (B < 0 || N < 1 || H < 1 || W != 2 * H)
    return eml::fail(EML_EINVAL, "eml_sg_rasterise_f32: need N>=1, H>=1, W==2H (got B=%d N=%d H=%d W=%d)", B, N, H, W);
  if (flags & ~EML_SG_EXHAUSTIVE) return eml::fail(EML_EINVAL, "eml_sg_rasterise_ex_f32: unknown flags 0x%x", flags);
  if (B == 0) return EML_OK;
  if (B > 65535) return eml::fail(EML_EINVAL, "eml_sg_rasterise_f32: B=%d exceeds grid.z", B);
  const float step = (float)(3.14159265358979323846 / (double)H);
  dim3 grid((W + kTileW - 1) / kTileW, (H + kTileH - 1) / kTileH, B);
  const hipStream_t st = (hipStream_t)stream;
  if (flags & EML_SG_EXHAUSTIVE) {
    if (executed_exp)
      hipLaunchKernelGGL((sg_rasterise_kernel<true, true>), grid, dim3(256), 0, st, dirs, sizes, colors, out, N, H, W, step, executed_exp);
    else
      hipLaunchKernelGGL((sg_rasterise_kernel<true, false>), grid, dim3(256), 0, st, dirs, sizes, colors, out, N, H, W, step, nullptr);
  } else if (executed_exp) {
    hipLaunchKernelGGL((sg_rasterise_kernel<false, true>), grid, dim3(256), 0, st, dirs, sizes, colors, out, N, H, W, step, executed_exp);
  } else {
    hipLaunchKernelGGL((sg_rasterise_kernel<false, false>), grid, dim3(256), 0, st, dirs, sizes, colors, out, N, H, W, step, nullptr);
  }
  return eml::check_launch("eml_sg_rasterise_f32");
}

extern "C" int eml_sg_rasterise_f32(const float* dirs, const float* sizes, const float* colors,
                                    float* out, int B, int N, int H, int W, eml_stream_t stream) {
  return eml_sg_rasterise_ex_f32(dirs, sizes, colors, out, B, N, H, W, 0, nullptr, stream);
}

extern "C" int eml_sg_rasterise_bwd_colors_f32(const float* dirs, const float* sizes,
                                               const float* gout, float* gcolors, int B, int N,
                                               int H, int W, eml_stream_t stream) {
  if (!dirs || !sizes || !gout || !gcolors)
    return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_f32: null pointer");
  if (B < 0 || N < 1 || H < 1 || W != 2 * H)
    return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_f32: need N>=1, H>=1, W==2H");
  if (B == 0) return EML_OK;
  if (B > 65535) return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_f32: B=%d exceeds grid.y", B);
  const float step = (float)(3.14159265358979323846 / (double)H);
  const size_t lds = (size_t)(2 * H + 2 * W + 4 * kLG * 3) * sizeof(float);
  dim3 grid((N + kLG - 1) / kLG, B);
  hipLaunchKernelGGL(sg_rasterise_bwd_colors_kernel, grid, dim3(256), lds, (hipStream_t)stream, dirs,
                     sizes, gout, gcolors, N, H, W, step);
  return eml::check_launch("eml_sg_rasterise_bwd_colors_f32");
}

// The culled form (see sg_rasterise_bwd_colors_tiled_kernel).  work: eml_sg_rasterise_bwd_work_floats(B, N, H, W) floats.
// flags: EML_SG_EXHAUSTIVE = every light for every tile (the reference sums; what the culled launch must equal bit for bit).
extern "C" size_t eml_sg_rasterise_bwd_work_floats(int B, int N, int H, int W) {
  if (B < 1 || N < 1 || H < 1 || W < 1) return 0;
  return (size_t)B * ((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH) * N * 3;
}

extern "C" int eml_sg_rasterise_bwd_colors_ex_f32(const float* dirs, const float* sizes, const float* gout, float* gcolors,
                                                  float* work, int B, int N, int H, int W, int flags, eml_stream_t stream) {
  if (!dirs || !sizes || !gout || !gcolors || !work)
    return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_ex_f32: null pointer");
  if (B < 0 || N < 1 || H < 1 || W != 2 * H)
    return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_ex_f32: need N>=1, H>=1, W==2H");
  if (flags & ~EML_SG_EXHAUSTIVE) return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_ex_f32: unknown flags 0x%x", flags);
  if (B == 0) return EML_OK;
  if (B > 65535) return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_ex_f32: B=%d exceeds grid.z", B);
  if ((long)B * N * 3 > 2147483647L) return eml::fail(EML_EINVAL, "eml_sg_rasterise_bwd_colors_ex_f32: too many lights");
  const float step = (float)(3.14159265358979323846 / (double)H);
  const dim3 grid((W + kTileW - 1) / kTileW, (H + kTileH - 1) / kTileH, B);
  const hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sg_rasterise_bwd_colors_tiled_kernel, grid, dim3(256), 0, st, dirs, sizes, gout, work, N, H, W, step,
                     (flags & EML_SG_EXHAUSTIVE) ? 1 : 0);
  int rc = eml::check_launch("eml_sg_rasterise_bwd_colors_ex_f32");
  if (rc) return rc;
  const int total = B * N * 3;
  hipLaunchKernelGGL(sg_reduce_tiles_kernel, dim3((total + 255) / 256), dim3(256), 0, st, work, (int)(grid.x * grid.y), N * 3,
                     gcolors, total);
  return eml::check_launch("eml_sg_rasterise_bwd_colors_ex_f32(reduce)");
}
