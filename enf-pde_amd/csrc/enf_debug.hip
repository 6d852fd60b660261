// enf_debug.hip -- layout self-test: Y (M x 32) = A (M x K) . X (K x 32) through the same
// pack_panel + make_frags + gemm_stage path the production kernels use.  Test-only entry point.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_device.h"

template <int KBIN, int MBOUT, bool BF16>
__global__ __launch_bounds__(64) void enf_debug_gemm_kernel(const char* panel, const float* X, float* Y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  constexpr int BYTES = KBIN * MBOUT * (BF16 ? 2048 : 4096);
  for (int i = lane; i < BYTES / 16; i += 64)
    reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(panel)[i];
  __syncthreads();
  f32x16 x[KBIN];
#pragma unroll
  for (int k = 0; k < KBIN; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[k][r] = X[(32 * k + RHO(r, 0) + 4 * half) * 32 + col];
  Frags<BF16, KBIN> F;
  make_frags<BF16, KBIN>(F, x);
  f32x16 acc[MBOUT];
#pragma unroll
  for (int m = 0; m < MBOUT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  gemm_stage<BF16, KBIN, MBOUT>(acc, F, smem, lane);
#pragma unroll
  for (int m = 0; m < MBOUT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[(32 * m + RHO(r, 0) + 4 * half) * 32 + col] = acc[m][r];
}

// panel: packed A (M x K) in the requested precision; X: K x 32 row-major; Y: M x 32 row-major
extern "C" int enf_debug_gemm(const void* panel, const float* X, float* Y, int M, int K, int bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)(M / 32) * (K / 32) * (bf16 ? 2048 : 4096);
#define C(KB, MB)                                                                                                   \
  if (K == 32 * KB && M == 32 * MB) {                                                                               \
    if (bf16) hipLaunchKernelGGL((enf_debug_gemm_kernel<KB, MB, true>), dim3(1), dim3(64), bytes, st, (const char*)panel, X, Y);  \
    else hipLaunchKernelGGL((enf_debug_gemm_kernel<KB, MB, false>), dim3(1), dim3(64), bytes, st, (const char*)panel, X, Y);      \
    return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;                                                        \
  }
  C(2, 2) C(4, 2) C(1, 2)
#undef C
  return ENF_EUNSUPPORTED;
}
