// Read-pattern probe: the first `ncols` columns of a (P, LD) f32 matrix read (a) as the 1x1 forward's MFMA operand loads read
// them -- a wave owns 16 rows, one instruction loads 16 rows x 64 bytes (lane (r, kk): row r, columns 16 j + 4 kk) --, (b) 8
// consecutive lanes per row = one whole line, 8 rows per instruction, (c) row by row, consecutive lanes consecutive 16 bytes,
// (d) as the 1x1 weight gradient reads them: float2 per lane, lane i of a 32-channel group: 4 rows x 128 bytes per instruction.
// Every variant sums what it reads into a per-lane accumulator (written once at the end) so that nothing is optimised away.
#include <hip/hip_runtime.h>
extern "C" {
__global__ __launch_bounds__(256) void rp_dt(const float* G, long P, int LD, int ncols, float* out) {
  const int lane = threadIdx.x & 63, r = lane & 15, kk = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long t = wave; t * 16 < P; t += nw) {
    const long row = min(t * 16 + r, P - 1);
    for (int j = 0; j < ncols / 16; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(G + row * LD + 16 * j + 4 * kk);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = a.x + a.y + a.z + a.w;
}
__global__ __launch_bounds__(256) void rp_dt8(const float* G, long P, int LD, int ncols, float* out) {
  const int lane = threadIdx.x & 63, r = lane >> 3, k8 = lane & 7;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long t = wave; t * 16 < P; t += nw)
    for (int h = 0; h < 2; ++h) {
      const long row = min(t * 16 + 8 * h + r, P - 1);
      for (int j = 0; j < ncols / 32; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(G + row * LD + 32 * j + 4 * k8);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = a.x + a.y + a.z + a.w;
}
__global__ __launch_bounds__(256) void rp_rows(const float* G, long P, int LD, int ncols, float* out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  const int q = ncols / 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long t = wave; t * 16 < P; t += nw)
    for (int e = lane; e < 16 * q; e += 64) {
      const long row = min(t * 16 + e / q, P - 1);
      const float4 v = *reinterpret_cast<const float4*>(G + row * LD + 4 * (e % q));
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = a.x + a.y + a.z + a.w;
}
__global__ __launch_bounds__(256) void rp_f2(const float* G, long P, int LD, int ncols, float* out) {
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float2 a = make_float2(0.f, 0.f);
  for (long t = wave; t * 4 < P; t += nw) {
    const long row = min(t * 4 + kk, P - 1);
    for (int g = 0; g < ncols / 32; ++g) {
      const float2 v = *reinterpret_cast<const float2*>(G + row * LD + 32 * g + 2 * i);
      a.x += v.x; a.y += v.y;
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = a.x + a.y;
}
int rp_run(int which, const float* G, long P, int LD, int ncols, int grid, float* out, void* stream) {
  if (which == 0) hipLaunchKernelGGL(rp_dt, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols, out);
  else if (which == 1) hipLaunchKernelGGL(rp_dt8, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols, out);
  else if (which == 2) hipLaunchKernelGGL(rp_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols, out);
  else hipLaunchKernelGGL(rp_f2, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols, out);
  return (int)hipGetLastError();
}
}
