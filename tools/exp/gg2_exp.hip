// Experiment harness (NOT product): several configurations of gather_gemm2_kernel behind one C entry point, for
// tools/exp/gg2_bench.py (A/B against the round-2 kernel of libemlight_hip.so on the projector's layer shapes).
#include <cstdlib>

#include "gather_gemm3_exp.h"

namespace {
template <int BN, int NT, int LPP, bool PK, bool ONE>
int launch(const float* X, const int* idx, const float* wgt, const float* W2, const float* bias, float* Y, long M, int HW,
           int Po, int C, int O, int ke, const unsigned char* rowmax, const float* res, float slope, hipStream_t stream) {
  size_t lds = gg2::lds_bytes<BN, NT>();
  if (getenv("GG2_LDS_PAD")) lds += (size_t)atoi(getenv("GG2_LDS_PAD"));   // experiment: force one workgroup per CU
  const long n_mt = (M + gg2::kBM - 1) / gg2::kBM, per_xcd = (n_mt + 7) / 8;
  const dim3 grid((unsigned)(8 * per_xcd * (O / BN)));
  auto kern = gg2::gather_gemm2_kernel<BN, NT, LPP, PK, ONE>;
  EML_ENSURE_LDS(kern, lds);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, X, idx, wgt, W2, bias, Y, (int)M, HW, Po, C, O, ke, rowmax, res, slope,
                     gg2::SpadeEpilogue{});
  return eml::check_launch("gg2_exp");
}
template <int BN, int LPP, bool ONE>
int launch3(const float* X, const int* idx, const float* wgt, const float* W2, const float* bias, float* Y, long M, int HW,
            int Po, int C, int O, int ke, const unsigned char* rowmax, const float* res, float slope, hipStream_t stream) {
  size_t lds = gg2::lds_bytes<BN, 512>();
  if (getenv("GG2_LDS_PAD")) lds += (size_t)atoi(getenv("GG2_LDS_PAD"));
  const long n_mt = (M + gg2::kBM - 1) / gg2::kBM, per_xcd = (n_mt + 7) / 8;
  const dim3 grid((unsigned)(8 * per_xcd * (O / BN)));
  auto kern = gg2::gather_gemm3_kernel<BN, LPP, ONE>;
  EML_ENSURE_LDS(kern, lds);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, X, idx, wgt, W2, bias, Y, (int)M, HW, Po, C, O, ke, rowmax, res, slope);
  return eml::check_launch("gg3_exp");
}
}  // namespace

// variant: 0 = LPP 8, scalar combine; 1 = LPP 8, packed combine; 2 = LPP 4, scalar; 3 = BN 256 (512 threads) where O % 256 == 0;
// 4 = 512 threads on the 128 x {128, 64} tile (wave tile 64 x {32, 16})
extern "C" int gg2_exp_fwd(int variant, const float* X, const int* idx, const float* wgt, const float* W2, const float* bias,
                           float* Y, int B, int HW, int Po, int C, int O, int ke, const unsigned char* rowmax,
                           const float* res, float slope, void* stream) {
  const long M = (long)B * Po;
  hipStream_t st = (hipStream_t)stream;
  if (C < 64 || C % 32 || O % 64) return eml::fail(EML_EINVAL, "gg2_exp_fwd: C >= 64, C %% 32 == 0, O %% 64 == 0");
#define GO3(BN, LPP, ONE) return launch3<BN, LPP, ONE>(X, idx, wgt, W2, bias, Y, M, HW, Po, C, O, ke, rowmax, res, slope, st)
  if (variant == 5 || variant == 6) {   // wave-specialised: 5 = full-line gathers, 6 = half-line
    if (ke == 1) { if (O % 128 == 0) GO3(128, 8, true); GO3(64, 8, true); }
    if (variant == 5) { if (O % 128 == 0) GO3(128, 8, false); GO3(64, 8, false); }
    if (O % 128 == 0) GO3(128, 4, false); GO3(64, 4, false);
  }
#define GO(BN, NT, LPP, PK, ONE) return launch<BN, NT, LPP, PK, ONE>(X, idx, wgt, W2, bias, Y, M, HW, Po, C, O, ke, rowmax, res, slope, st)
  if (ke == 1) {
    if (variant == 3 && O % 256 == 0) GO(256, 512, 8, false, true);
    if (O % 128 == 0) GO(128, 256, 8, false, true);
    GO(64, 256, 8, false, true);
  }
  if (variant == 3 && O % 256 == 0) GO(256, 512, 8, false, false);
  if (variant == 4 && O % 128 == 0) GO(128, 512, 8, false, false);   // 8 waves of 64 x 32: four waves per SIMD
  if (variant == 4) GO(64, 512, 8, false, false);
  if (O % 128 == 0) {
    if (variant == 1) GO(128, 256, 8, true, false);
    if (variant == 2) GO(128, 256, 4, false, false);
    GO(128, 256, 8, false, false);
  }
  if (variant == 1) GO(64, 256, 8, true, false);
  if (variant == 2) GO(64, 256, 4, false, false);
  GO(64, 256, 8, false, false);
#undef GO
}
