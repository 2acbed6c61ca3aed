// Spherical-Gaussian lobes -> equirectangular panorama, forward and d/d(colors).
//
// Replaces the per-light Python loop of convert_to_panorama
// (reference RegressionNetwork/util.py:222-245): there each of the N lights costs ~6 ATen
// launches and a full re-read/re-write of the (B,3,H,W) accumulator.  Here one launch keeps
// the three channel accumulators of a pixel in registers, streams the lights of the sample
// through LDS (broadcast reads), and writes the panorama exactly once.
//
// Roofline: B*N*H*W exp2 evaluations against B*3*H*W*4 output bytes.  At N=128 that is
// 128 transcendentals per 12 output bytes -> transcendental-(VALU-)bound, not HBM-bound;
// the wave-uniform cull below removes the lights whose lobe underflows to exactly 0 for
// all 64 pixels of a wave's 16x4 tile (exp2(t) == 0 for t < -150 in f32), which is
// bit-identical to adding them.
#include "eml_common.h"

namespace {

constexpr int kTileW = 32;  // block tile: 2x2 waves, each wave a 16(w) x 4(h) pixel patch
constexpr int kTileH = 8;
constexpr int kChunk = 512;  // lights staged per LDS pass (8 floats each = 16 KiB)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kCull = -150.0f;  // exp2(-150) underflows below the smallest f32 denormal

struct __attribute__((aligned(16))) Lobe {
  float dx, dy, dz, k2;  // direction, log2(e)/size
  float r, g, b, pad;    // colour
};

// f32 view vector of pixel (h, w) on the H x 2H grid, evaluated like the reference
// (util.py:223-233): theta = (h+.5)*f32(pi/H), phi = (w+.5)*f32(pi/H).
__device__ __forceinline__ void view_vector(int h, int w, float step, float& x, float& y, float& z) {
  const float th = ((float)h + 0.5f) * step;
  const float ph = ((float)w + 0.5f) * step;
  float st, ct, sp, cp;
  sincosf(th, &st, &ct);
  sincosf(ph, &sp, &cp);
  x = st * cp;
  y = st * sp;
  z = ct;
}

__global__ __launch_bounds__(256) void sg_rasterise_kernel(
    const float* __restrict__ dirs, const float* __restrict__ sizes,
    const float* __restrict__ colors, float* __restrict__ out, int N, int H, int W, float step) {
  __shared__ Lobe lobes[kChunk];
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int w = blockIdx.x * kTileW + (wave & 1) * 16 + (lane & 15);
  const int h = blockIdx.y * kTileH + (wave >> 1) * 4 + (lane >> 4);
  float px, py, pz;
  view_vector(h, w, step, px, py, pz);

  float ar = 0.f, ag = 0.f, ab = 0.f;
  for (int base = 0; base < N; base += kChunk) {
    const int cnt = min(kChunk, N - base);
    __syncthreads();
    for (int i = tid; i < cnt; i += 256) {
      const size_t li = (size_t)b * N + base + i;
      Lobe L;
      L.dx = dirs[3 * li + 0];
      L.dy = dirs[3 * li + 1];
      L.dz = dirs[3 * li + 2];
      L.k2 = kLog2e / sizes[li];
      L.r = colors[3 * li + 0];
      L.g = colors[3 * li + 1];
      L.b = colors[3 * li + 2];
      L.pad = 0.f;
      lobes[i] = L;
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      const Lobe L = lobes[i];  // same address on every lane: LDS broadcast
      const float dot = fmaf(L.dz, pz, fmaf(L.dy, py, L.dx * px));
      const float t = (dot - 1.0f) * L.k2;
      if (__builtin_amdgcn_ballot_w64(t > kCull) == 0) continue;  // wave-uniform, exact
      const float e = __builtin_amdgcn_exp2f(t);
      ar = fmaf(L.r, e, ar);
      ag = fmaf(L.g, e, ag);
      ab = fmaf(L.b, e, ab);
    }
  }
  if (h < H && w < W) {
    const size_t plane = (size_t)H * W;
    float* o = out + (size_t)b * 3 * plane + (size_t)h * W + w;
    o[0] = ar;
    o[plane] = ag;
    o[2 * plane] = ab;
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

}  // namespace

extern "C" int eml_sg_rasterise_f32(const float* dirs, const float* sizes, const float* colors,
                                    float* out, int B, int N, int H, int W, eml_stream_t stream) {
  if (!dirs || !sizes || !colors || !out) return eml::fail(EML_EINVAL, "eml_sg_rasterise_f32: null pointer");
  if (B < 0 || N < 1 || H < 1 || W != 2 * H)
    return eml::fail(EML_EINVAL, "eml_sg_rasterise_f32: need N>=1, H>=1, W==2H (got B=%d N=%d H=%d W=%d)", B, N, H, W);
  if (B == 0) return EML_OK;
  if (B > 65535) return eml::fail(EML_EINVAL, "eml_sg_rasterise_f32: B=%d exceeds grid.z", B);
  const float step = (float)(3.14159265358979323846 / (double)H);
  dim3 grid((W + kTileW - 1) / kTileW, (H + kTileH - 1) / kTileH, B);
  hipLaunchKernelGGL(sg_rasterise_kernel, grid, dim3(256), 0, (hipStream_t)stream, dirs, sizes, colors,
                     out, N, H, W, step);
  return eml::check_launch("eml_sg_rasterise_f32");
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
