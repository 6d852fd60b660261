// enf_loss.hip -- the inner loop's reconstruction loss and its gradient in one pass.
//   loss = mean((out - target)^2)          (pde_trainer.py:185)
//   dout = 2 (out - target) / n * grad_scale
// so that a fit step is forward -> this kernel -> backward, without a framework autograd graph of tiny
// elementwise kernels in between.  `loss` is accumulated with one atomic per block: the caller zeroes it.
#include <hip/hip_runtime.h>
#include "enf_layout.h"

__global__ __launch_bounds__(256) void enf_mse_kernel(const float* __restrict__ out, const float* __restrict__ target, size_t n,
                                                      float inv_n, float gscale, float* __restrict__ dout, float* loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = out[i] - target[i];
    s = fmaf(d, d, s);
    if (dout) dout[i] = 2.0f * d * inv_n * gscale;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}

extern "C" int enf_mse_value_grad(const float* out, const float* target, size_t n, float grad_scale, float* dout, float* loss,
                                  void* stream) {
  if (!out || !target || !loss || n == 0) return ENF_EINVAL;
  size_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(enf_mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, target, n, 1.0f / (float)n,
                     grad_scale, dout, loss);
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}
