// Write-pattern probe: the same (P, LD) f32 matrix written (a) as MFMA D^T epilogues write it -- a wave owns 16 rows, one
// instruction stores 16 rows x 64 bytes (lane (r, kk): row r, columns 16 nt + 4 kk .. + 3) -- and (b) row-linear, one instruction
// = 1 KB contiguous; (c) as (a) but two column tiles back to back per row visit... all values a function of the address.
#include <hip/hip_runtime.h>
extern "C" {
__global__ __launch_bounds__(256) void wp_dt(float* G, long P, int LD, int ncols) {
  const int lane = threadIdx.x & 63, r = lane & 15, kk = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long t = wave; t * 16 < P; t += nw) {
    const long row = min(t * 16 + r, P - 1);
    for (int nt = 0; nt < ncols / 16; ++nt)
      *reinterpret_cast<float4*>(G + row * LD + 16 * nt + 4 * kk) = make_float4((float)nt, (float)r, 1.f, 2.f);
  }
}
__global__ __launch_bounds__(256) void wp_rows(float* G, long P, int LD, int ncols) {
  // one row = ncols floats; lanes walk the row: 8 lanes x 16 B = one 128-byte line, a wave instruction = 8 lines of ONE row pair
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  const int q = ncols / 4;                    // float4 per row
  for (long t = wave; t * 16 < P; t += nw) {
    for (int e = lane; e < 16 * q; e += 64) {
      const long row = min(t * 16 + e / q, P - 1);
      const int c4 = e % q;
      *reinterpret_cast<float4*>(G + row * LD + 4 * c4) = make_float4((float)c4, 3.f, 1.f, 2.f);
    }
  }
}
// (d) the D^T tile after a lane permutation: lane l -> row l >> 2, piece l & 3: 4 CONSECUTIVE lanes write the 64 bytes of a row
__global__ __launch_bounds__(256) void wp_dt4(float* G, long P, int LD, int ncols) {
  const int lane = threadIdx.x & 63, r = lane >> 2, kk = lane & 3;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long t = wave; t * 16 < P; t += nw) {
    const long row = min(t * 16 + r, P - 1);
    for (int nt = 0; nt < ncols / 16; ++nt)
      *reinterpret_cast<float4*>(G + row * LD + 16 * nt + 4 * kk) = make_float4((float)nt, (float)r, 1.f, 2.f);
  }
}
// (e) two column tiles per instruction: lane l -> row l >> 3, piece l & 7: 8 consecutive lanes write one 128-byte line, 8 rows
__global__ __launch_bounds__(256) void wp_dt8(float* G, long P, int LD, int ncols) {
  const int lane = threadIdx.x & 63, r = lane >> 3, k8 = lane & 7;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long t = wave; t * 16 < P; t += nw) {
    for (int h = 0; h < 2; ++h) {
      const long row = min(t * 16 + 8 * h + r, P - 1);
      for (int nt = 0; nt < ncols / 32; ++nt)
        *reinterpret_cast<float4*>(G + row * LD + 32 * nt + 4 * k8) = make_float4((float)nt, (float)r, 1.f, 2.f);
    }
  }
}
__global__ __launch_bounds__(256) void wp_flat(float* G, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    reinterpret_cast<float4*>(G)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int wp_run(int which, float* G, long P, int LD, int ncols, int grid, void* stream) {
  if (which == 0) hipLaunchKernelGGL(wp_dt, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols);
  else if (which == 1) hipLaunchKernelGGL(wp_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols);
  else if (which == 3) hipLaunchKernelGGL(wp_dt4, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols);
  else if (which == 4) hipLaunchKernelGGL(wp_dt8, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P, LD, ncols);
  else hipLaunchKernelGGL(wp_flat, dim3(grid), dim3(256), 0, (hipStream_t)stream, G, P * LD / 4);
  return (int)hipGetLastError();
}
}
