// enf_debug.hip -- layout self-test: Y (M x 16) = A (M x K) . X (K x 16) through the same
// pack_panel + make_frags + gemm_stage path the production kernels use.  Test-only entry point.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_device.h"

template <int KBIN, int MTOUT, bool BF16>
__global__ __launch_bounds__(64) void enf_debug_gemm_kernel(const char* panel, const float* X, float* Y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, col = lane & 15, quad = lane >> 4;
  constexpr int BYTES = PanelCfg<KBIN, MTOUT, BF16>::BYTES;
  for (int i = lane; i < BYTES / 16; i += 64)
    reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(panel)[i];
  __syncthreads();
  f32x4 x[2 * KBIN];
#pragma unroll
  for (int t = 0; t < 2 * KBIN; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) x[t][i] = X[(16 * t + 4 * quad + i) * 16 + col];
  Frags<BF16, KBIN> F;
  make_frags<BF16, KBIN>(F, x);
  f32x4 acc[MTOUT];
#pragma unroll
  for (int m = 0; m < MTOUT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_stage<BF16, KBIN, MTOUT>(acc, F, smem, lane);
#pragma unroll
  for (int m = 0; m < MTOUT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) Y[(16 * m + 4 * quad + i) * 16 + col] = acc[m][i];
}

// panel: packed A (M x K) in the requested precision; X: K x 16 row-major; Y: M x 16 row-major
extern "C" int enf_debug_gemm(const void* panel, const float* X, float* Y, int M, int K, int bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)M * K * (bf16 ? 2 : 4);
#define C(KB, MT)                                                                                                   \
  if (K == 32 * KB && M == 16 * MT) {                                                                               \
    if (bf16) hipLaunchKernelGGL((enf_debug_gemm_kernel<KB, MT, true>), dim3(1), dim3(64), bytes, st, (const char*)panel, X, Y);  \
    else hipLaunchKernelGGL((enf_debug_gemm_kernel<KB, MT, false>), dim3(1), dim3(64), bytes, st, (const char*)panel, X, Y);      \
    return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;                                                        \
  }
  C(2, 4) C(4, 4) C(1, 4) C(2, 1)
#undef C
  return ENF_EUNSUPPORTED;
}
