// Access-shape probe for the wide NHWC rows of the dense-block buffers (block 1: P = 4.9 M pixels, 224 floats = 896 B per
// row, a layer touches channels [0, k)).  Is the fabric rate of the 1x1 kernels (3.5-4.2 TB/s) a property of the MFMA
// operand access shape -- a wave instruction = 16 rows x 64-byte segments -- or of the data volume?
//   mode 0  "operand": lane (r = lane & 15, kk = lane >> 4) loads float4 at row 16*t + r, column 16*j + 4*kk, j = 0..k/16-1
//           (exactly the forward / dgrad kernels' pattern, 2 row tiles in flight per wave)
//   mode 1  "rows":    lane i loads float4 at row p, column 4*i (i < k/4): one whole row per wave instruction (sequential bursts)
//   mode 2/3: the same two shapes as read-modify-write (G += 1)
//   mode 4  "wgrad":   conv1x1_bwd_weight's x operand (round 4): lane (r, kk) loads float2 at pixel 4q + kk,
//           channels 32 * (wave + 4i) + 2r -- a wave instruction = 4 rows x 128 B, 8 bytes per lane; 4 pixel quads in flight
//   mode 5  "wgrad4":  the same rows as float4: channels 64 * (wave + 4i) + 4r -- 4 rows x 256 B, 16 bytes per lane
// Persistent grid, 256 threads; prints nothing: timed from the host with HIP events.  Build:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/exp/row_access_probe.hip -o build_exp/row_probe.so
#include <hip/hip_runtime.h>

extern "C" __global__ __launch_bounds__(256, 2) void probe_kernel(float* __restrict__ X, int ld, long P, int k, int mode,
                                                                    float* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc = 0.f;
  if (mode == 0 || mode == 2) {
    const int r = lane & 15, kk = lane >> 4;
    const long ntiles = (P + 127) / 128;   // 4 waves x 32 pixels
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long p0 = tile * 128 + wave * 32;
      const long pa = min(p0 + r, P - 1), pb = min(p0 + 16 + r, P - 1);
      for (int j = 0; j < k / 16; j += 2) {
        float4 v[4];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int c = 16 * min(j + n, k / 16 - 1) + 4 * kk;
          v[2 * n] = *reinterpret_cast<const float4*>(X + pa * ld + c);
          v[2 * n + 1] = *reinterpret_cast<const float4*>(X + pb * ld + c);
        }
        if (mode == 2) {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int c = 16 * min(j + n, k / 16 - 1) + 4 * kk;
            float4 a = v[2 * n], b = v[2 * n + 1];
            a.x += 1.f, a.y += 1.f, a.z += 1.f, a.w += 1.f;
            b.x += 1.f, b.y += 1.f, b.z += 1.f, b.w += 1.f;
            *reinterpret_cast<float4*>(X + pa * ld + c) = a;
            *reinterpret_cast<float4*>(X + pb * ld + c) = b;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
        }
      }
    }
  } else if (mode == 4 || mode == 5) {
    const int r = lane & 15, kk = lane >> 4;
    const int gw = mode == 4 ? 32 : 64;                 // channels per group
    const int ngroups = (k + gw - 1) / gw;
    const long nchunks = (P + 63) / 64;
    for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
      for (int q0 = 0; q0 < 16; q0 += 4) {
        if (mode == 4) {
          float2 v[4][3];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const long pc = min(chunk * 64 + 4 * (q0 + u) + kk, P - 1);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const int g = wave + 4 * i, c = min(gw * g + 2 * r, k - 2);
              v[u][i] = g < ngroups ? *reinterpret_cast<const float2*>(X + pc * ld + c) : make_float2(0.f, 0.f);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc += v[u][i].x + v[u][i].y;
        } else {
          float4 v[4][2];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const long pc = min(chunk * 64 + 4 * (q0 + u) + kk, P - 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int g = wave + 4 * i, c = min(gw * g + 4 * r, k - 4);
              v[u][i] = g < ngroups ? *reinterpret_cast<const float4*>(X + pc * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc += v[u][i].x + v[u][i].y + v[u][i].z + v[u][i].w;
        }
      }
    }
  } else {
    const bool on = 4 * lane < k;
    const int c = on ? 4 * lane : 0;
    const long nrows4 = (P + 15) / 16;     // 4 waves x 4 rows in flight
    for (long t = blockIdx.x; t < nrows4; t += gridDim.x) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long p = min(t * 16 + wave * 4 + q, P - 1);
        v[q] = *reinterpret_cast<const float4*>(X + p * ld + c);
      }
      if (mode == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long p = min(t * 16 + wave * 4 + q, P - 1);
          float4 a = v[q];
          a.x += 1.f, a.y += 1.f, a.z += 1.f, a.w += 1.f;
          if (on) *reinterpret_cast<float4*>(X + p * ld + c) = a;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += on ? v[q].x + v[q].y + v[q].z + v[q].w : 0.f;
      }
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

extern "C" int probe_launch(float* X, int ld, long P, int k, int mode, int grid, float* sink, void* stream) {
  hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, ld, P, k, mode, sink);
  return (int)hipGetLastError();
}
