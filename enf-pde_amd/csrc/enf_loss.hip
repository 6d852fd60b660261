// enf_loss.hip -- the inner loop's reconstruction loss and its gradient in one pass.
//   loss = mean((out - target)^2)          (pde_trainer.py:185)
//   dout = 2 (out - target) / n * grad_scale
// so that a fit step is forward -> this kernel -> backward, without a framework autograd graph of tiny
// elementwise kernels in between.  `loss` is accumulated with one atomic per block: the caller zeroes it.
//
// enf_meta_sgd_update: the meta-SGD update of every latent component in ONE launch (pde_trainer.py:206-219):
//   out_k = x_k - lr_k (scale g_k),   scale = the batch size (the gradient of a batch-mean loss, :206)
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

struct SgdArgs { EnfSgdSegment seg[ENF_SGD_MAX_SEGMENTS]; int nseg; float scale; };

__global__ __launch_bounds__(256) void enf_meta_sgd_kernel(SgdArgs A) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < ENF_SGD_MAX_SEGMENTS; ++k) {
    if (k >= A.nseg) return;
    const EnfSgdSegment& S = A.seg[k];
    if (i < S.n) {
      const int64_t row = i / S.width;
      const int c = (int)(i - row * S.width);
      const float g = S.g[row * S.g_stride + c] * A.scale;          // :206
      S.out[i] = S.x[i] - S.lr[S.lr_len == 1 ? 0 : c] * g;         // :215-219
      return;
    }
    i -= S.n;
  }
}

extern "C" int enf_meta_sgd_update(int nseg, const EnfSgdSegment* segs, float scale, void* stream) {
  if (nseg < 1 || nseg > ENF_SGD_MAX_SEGMENTS || !segs) return ENF_EINVAL;
  SgdArgs A{};
  int64_t total = 0;
  for (int k = 0; k < nseg; ++k) {
    const EnfSgdSegment& S = segs[k];
    if (!S.x || !S.g || !S.lr || !S.out || S.n <= 0 || S.width <= 0 || S.g_stride < S.width || S.n % S.width != 0) return ENF_EINVAL;
    if (S.lr_len != 1 && S.lr_len != S.width) return ENF_EDIM;
    A.seg[k] = S;
    total += S.n;
  }
  A.nseg = nseg;
  A.scale = scale;
  hipLaunchKernelGGL(enf_meta_sgd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

// enf_fit_inputs: the inner loop's setup in one launch (include/enf_hip.h) -- broadcast of the latent initialisation over the signals,
// gather of the S + 1 sampled coordinate / target sets, zeroed loss accumulators.  One flat index space over the five outputs.
struct FitInArgs {
  EnfFitComponent comp[ENF_SGD_MAX_SEGMENTS];
  int ncomp, B, Z, N, Ns, S1, dx, O;
  const float* coords; const float* img; const int64_t* masks;
  float* xs; float* ys; float* losses;
};

__global__ __launch_bounds__(256) void enf_fit_inputs_kernel(FitInArgs A) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < ENF_SGD_MAX_SEGMENTS; ++k) {
    if (k < A.ncomp) {
      const int64_t per = (int64_t)A.Z * A.comp[k].width, n = per * A.B;
      if (i < n) { A.comp[k].dst[i] = A.comp[k].src[i % per]; return; }
      i -= n;
    }
  }
  const int64_t nx = (int64_t)A.S1 * A.Ns * A.dx;
  if (i < nx) {                                   // xs[s][q][c] = coords[masks[q][s]][c]
    const int c = (int)(i % A.dx);
    const int64_t sq = i / A.dx, q = sq % A.Ns, s = sq / A.Ns;
    A.xs[i] = A.coords[A.masks[q * A.S1 + s] * A.dx + c];
    return;
  }
  i -= nx;
  const int64_t ny = (int64_t)A.S1 * A.B * A.Ns * A.O;
  if (i < ny) {                                   // ys[s][b][q][o] = img[b][masks[q][s]][o]
    const int o = (int)(i % A.O);
    int64_t r = i / A.O;
    const int64_t q = r % A.Ns; r /= A.Ns;
    const int64_t b = r % A.B, s = r / A.B;
    A.ys[i] = A.img[(b * A.N + A.masks[q * A.S1 + s]) * A.O + o];
    return;
  }
  i -= ny;
  if (i < A.S1) A.losses[i] = 0.f;
}

extern "C" int enf_fit_inputs(int ncomp, const EnfFitComponent* comps, int32_t B, int32_t Z, int32_t N, int32_t Ns, int32_t S1, int32_t dx,
                              int32_t O, const float* coords, const float* img, const int64_t* masks, float* xs, float* ys, float* losses,
                              void* stream) {
  if (ncomp < 1 || ncomp > ENF_SGD_MAX_SEGMENTS || !comps || !coords || !img || !masks || !xs || !ys || !losses) return ENF_EINVAL;
  if (B < 1 || Z < 1 || N < 1 || Ns < 1 || S1 < 1 || dx < 1 || O < 1) return ENF_EDIM;
  FitInArgs A{};
  int64_t total = 0;
  for (int k = 0; k < ncomp; ++k) {
    if (!comps[k].src || !comps[k].dst || comps[k].width < 1) return ENF_EINVAL;
    A.comp[k] = comps[k];
    total += (int64_t)B * Z * comps[k].width;
  }
  A.ncomp = ncomp; A.B = B; A.Z = Z; A.N = N; A.Ns = Ns; A.S1 = S1; A.dx = dx; A.O = O;
  A.coords = coords; A.img = img; A.masks = masks; A.xs = xs; A.ys = ys; A.losses = losses;
  total += (int64_t)S1 * Ns * dx + (int64_t)S1 * B * Ns * O + S1;
  hipLaunchKernelGGL(enf_fit_inputs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

