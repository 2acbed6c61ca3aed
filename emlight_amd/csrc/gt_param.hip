// Ground-truth parametrisation of HDR panoramas on the device (reference
// RegressionNetwork/representation/distribution_representation.py:65-120, `extract_mesh`): the inverse of the
// spherical-Gaussian rasteriser -- nearest-anchor Voronoi binning of a steradian-weighted panorama into per-anchor
// RGB energy plus the ambient remainder.  The reference is float64 numpy on one image at a time with a Python loop
// over the anchors (each iteration masks the whole panorama); here: a one-off index map, one max-reduction and one
// segmented (CSR) reduction per batch, all in f64, deterministic (fixed-order tree reductions, no atomics).
#include "eml_common.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

// numpy.linspace(0, stop, n)[i]: i * (stop / (n - 1)), last element exactly `stop`
__device__ __forceinline__ double linspace_at(int i, int n, double stop) {
  if (n <= 1) return 0.0;
  return (i == n - 1) ? stop : (double)i * (stop / (double)(n - 1));
}

// idx[h][w] = argmin_i || v(h, w) - a_i ||, v on the ENDPOINT-INCLUSIVE grid (:74-87)
__global__ __launch_bounds__(256) void gt_anchor_index_kernel(const double* __restrict__ anchors, int N, int H, int W,
                                                              int* __restrict__ idx) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int h = p / W, w = p - h * W;
  const double th = linspace_at(h, H, kPi), ph = linspace_at(w, W, 2.0 * kPi);
  const double vx = sin(th) * cos(ph), vy = sin(th) * sin(ph), vz = cos(th);
  double best = 1e300;
  int bi = 0;
  for (int i = 0; i < N; ++i) {
    const double dx = vx - anchors[3 * i], dy = vy - anchors[3 * i + 1], dz = vz - anchors[3 * i + 2];
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d < best) {
      best = d;
      bi = i;
    }
  }
  idx[p] = bi;
}

// steradian-weighted pixel (:92-93): hdr = sin((h + .5) / H * pi) * hdr, lum = .3 r + .59 g + .11 b, all f64
__device__ __forceinline__ double weighted(const float* __restrict__ px, int h, int H, double& r, double& g, double& b) {
  const double st = sin(((double)h + 0.5) / (double)H * kPi);
  r = st * (double)px[0];
  g = st * (double)px[1];
  b = st * (double)px[2];
  return 0.3 * r + 0.59 * g + 0.11 * b;
}

__device__ __forceinline__ double block_reduce(double v, double* red, bool is_max) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {  // fixed-order tree: deterministic
    if (tid < s) red[tid] = is_max ? fmax(red[tid], red[tid + s]) : red[tid] + red[tid + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

__global__ __launch_bounds__(1024) void gt_weighted_max_kernel(const float* __restrict__ hdr, int H, int W,
                                                               double* __restrict__ maxv) {
  __shared__ double red[1024];
  const float* img = hdr + (size_t)blockIdx.x * H * W * 3;
  double m = -1e300;
  for (int p = threadIdx.x; p < H * W; p += 1024) {
    double r, g, b;
    m = fmax(m, weighted(img + (size_t)p * 3, p / W, H, r, g, b));
  }
  m = block_reduce(m, red, true);
  if (threadIdx.x == 0) maxv[blockIdx.x] = m;
}

// block (a, b): a < N -> energy of anchor a (lit pixels of its Voronoi cell, :104-107); a == N -> ambient (:101) + map
__global__ __launch_bounds__(256) void gt_anchor_sums_kernel(const float* __restrict__ hdr, const int* __restrict__ cptr,
                                                             const int* __restrict__ cpix,
                                                             const double* __restrict__ maxv, int H, int W, int N,
                                                             double* __restrict__ sums, unsigned char* __restrict__ map) {
  __shared__ double red[256];
  const int a = blockIdx.x, b = blockIdx.y;
  const float* img = hdr + (size_t)b * H * W * 3;
  const double thr = maxv[b] * 0.05;
  double sr = 0.0, sg = 0.0, sb = 0.0;
  if (a < N) {
    for (int k = cptr[a] + threadIdx.x; k < cptr[a + 1]; k += 256) {
      const int p = cpix[k];
      double r, g, bl;
      if (weighted(img + (size_t)p * 3, p / W, H, r, g, bl) > thr) {
        sr += r;
        sg += g;
        sb += bl;
      }
    }
  } else {
    for (int p = threadIdx.x; p < H * W; p += 256) {
      double r, g, bl;
      const bool lit = weighted(img + (size_t)p * 3, p / W, H, r, g, bl) > thr;
      if (!lit) {
        sr += r;
        sg += g;
        sb += bl;
      }
      if (map) map[(size_t)b * H * W + p] = lit ? 1 : 0;
    }
  }
  sr = block_reduce(sr, red, false);
  sg = block_reduce(sg, red, false);
  sb = block_reduce(sb, red, false);
  if (threadIdx.x == 0) {
    double* o = sums + ((size_t)b * (N + 1) + a) * 3;
    o[0] = sr;
    o[1] = sg;
    o[2] = sb;
  }
}

}  // namespace

extern "C" int eml_gt_anchor_index_i32(const double* anchors, int N, int H, int W, int* idx, eml_stream_t stream) {
  if (!anchors || !idx || N < 1 || H < 1 || W < 1) return eml::fail(EML_EINVAL, "eml_gt_anchor_index_i32: bad arguments");
  hipLaunchKernelGGL(gt_anchor_index_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, anchors, N, H,
                     W, idx);
  return eml::check_launch("eml_gt_anchor_index_i32");
}

extern "C" int eml_gt_parametrise_f64(const float* hdr, const int* csr_ptr, const int* csr_pix, int B, int H, int W, int N,
                                      double* maxv, double* sums, unsigned char* map, eml_stream_t stream) {
  if (!hdr || !csr_ptr || !csr_pix || !maxv || !sums || B < 0 || H < 1 || W < 1 || N < 1 || B > 65535)
    return eml::fail(EML_EINVAL, "eml_gt_parametrise_f64: bad arguments");
  if (B == 0) return EML_OK;
  hipLaunchKernelGGL(gt_weighted_max_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, hdr, H, W, maxv);
  int rc = eml::check_launch("eml_gt_parametrise_f64(max)");
  if (rc) return rc;
  hipLaunchKernelGGL(gt_anchor_sums_kernel, dim3(N + 1, B), dim3(256), 0, (hipStream_t)stream, hdr, csr_ptr, csr_pix, maxv,
                     H, W, N, sums, map);
  return eml::check_launch("eml_gt_parametrise_f64(sums)");
}
