// Shared helpers for the gfx950 kernels of libemlight_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>

#include "../../include/emlight_hip.h"

namespace eml {

constexpr int kWave = 64;  // CDNA wavefront

inline char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// Call after a kernel launch: turns a launch error into a return code + message.
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fail(EML_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return (int)e > 0 ? (int)e : EML_ELAUNCH;
  }
  return EML_OK;
}

// Kernels that use more than the default 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised.
// That is a property of (kernel, device), not of a launch: it is set when a launch first needs more than any earlier one
// did (a grow-only high-water mark per device), not on every enqueue -- a training step made ~250 of these host calls.
constexpr int kMaxDevices = 64;
inline void ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<int>* high_water) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& hw = high_water[dev & (kMaxDevices - 1)];
  if ((int)bytes <= hw.load(std::memory_order_acquire)) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)bytes <= hw.load(std::memory_order_relaxed)) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  hw.store((int)bytes, std::memory_order_release);
}
#define EML_ENSURE_LDS(kernel_ptr, bytes)                                                     \
  do {                                                                                        \
    static std::atomic<int> eml_lds_hw_[eml::kMaxDevices];                                    \
    eml::ensure_dynamic_lds(reinterpret_cast<const void*>(kernel_ptr), (bytes), eml_lds_hw_); \
  } while (0)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- DPP cross-lane moves (VALU, no LDS round trip).  hipcc lowers __shfl_xor to ds_bpermute_b32 + an
// lgkmcnt(0) wait per step; inside an MFMA epilogue that serialises thousands of cycles (ISA of the
// conv1x1 dgrad: 32 dependent bpermutes per 16-channel tile).  Control words: quad_perm [1,0,3,2] = 0xB1,
// quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140 (a DPP "row" = 16 lanes).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }
// sum over the 16 lanes of a row (lanes with equal lane>>4); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// sum over the whole wave without LDS traffic (wave-uniform result): DPP inside the rows, the four row totals by v_readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum(v);
  const int b = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// min / max over the whole wave without LDS traffic: DPP inside the 16-lane rows, then the four row results through
// v_readlane (wave-uniform result).  The __shfl_xor forms above cost 6 ds_bpermute round trips each.
__device__ __forceinline__ float wave_min_dpp(float v) {
  v = fminf(v, dpp_mov<0xB1>(v));
  v = fminf(v, dpp_mov<0x4E>(v));
  v = fminf(v, dpp_mov<0x141>(v));
  v = fminf(v, dpp_mov<0x140>(v));
  const int b = __builtin_bit_cast(int, v);
  return fminf(fminf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
               fminf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const int b = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)),
                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences GLOBAL memory at workgroup
// scope, which the compiler implements as s_waitcnt vmcnt(0) before s_barrier: every barrier then waits for
// all of the wave's outstanding global loads AND stores (a tile's epilogue stores, a prefetch in flight).
// The persistent tile loops only hand LDS tiles between waves, so they use this one.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Tap-packed conv3x3 weights (dense_fwd_tp.hip): W2 [12][48][3][3] -> W2t[tile 7][J 3][lane 64][t 4].  lane = 16 kq + i; row i =
// 4 kkD + g of row tile `tile` is slot s = 4 tile + g of accumulator lane group kkD: tap = s / 3, o = 3 kkD + s % 3 (slot 27:
// zero); element e holds W2[o][16J + 4kq + t][tap].
__host__ __device__ __forceinline__ float w2t_value(const float* __restrict__ W2, int e) {
  const int t = e & 3, ln = (e >> 2) & 63, rest = e >> 8, J = rest % 3, tile = rest / 3;
  const int i = ln & 15, kq = ln >> 4, kkD = i >> 2, g = i & 3, s = 4 * tile + g;
  if (s >= 27) return 0.f;
  const int tap = s / 3, o = 3 * kkD + s % 3, c = 16 * J + 4 * kq + t;
  return W2[((size_t)o * 48 + c) * 9 + tap];
}
constexpr int kW2tFloats = 7 * 3 * 64 * 4;

}  // namespace eml
