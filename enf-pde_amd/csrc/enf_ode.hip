// enf_ode.hip -- the pair-wise part of the latent ODE's message passing.
//
// PonitaGen's ConvBlock (experiments/fitting/ode_models/ponita_ode_g.py:42-49) starts with the separable group
// convolution SepGconv (:63-83) over the fully connected latent set:
//     kernel[b,r,s,:] = kb[b,r,s,:] @ W            (B, Z, Z, C)   kb: the kernel basis (B, Z, Z, J)
//     out[b,r,:]      = bias + sum_s a[b,s,:] * kernel[b,r,s,:]
// The reference materialises `kernel`; here the J -> C product runs on the matrix pipe (fp32 16x16x4 MFMA, one
// 16-channel x 16-sender tile at a time) and is consumed in registers, so only kb, a and out touch memory.
//   forward          : out  (also d a: the same contraction with (r, s) swapped and the upstream gradient for a)
//   backward, basis  : d kb[b,r,s,:] = W (g[b,r,:] * a[b,s,:])
// d W = kb^T (g (x) a) over the pair axis is a plain GEMM and is left to the library (host side, like the decoder's
// per-pair weight gradients).  The sum over the MFMA's K index is order-free, so each lane's four K-steps use four
// CONSECUTIVE basis functions / channels (one 16-byte load feeds four MFMAs).
#include <hip/hip_runtime.h>
#include "enf_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct OdeConvArgs {
  const float* a;      // (B, Z, C)
  const float* kb;     // element (b, r, s, j) at b*Z*Z*J + r*sR + s*sS + j
  const float* W;      // (J, C)
  const float* bias;   // (C) or nullptr
  const float* g;      // (B, Z, C): upstream gradient (backward only)
  float* out;          // forward: (B, Z, C); backward: d kb (B, Z, Z, J)
  int B, Z, J, C;
  long sR, sS;
};

// one wave per (b, r, 16-channel tile); the 4 waves of a block take 4 channel tiles
template <int JM>   // J = 16 JM
__global__ __launch_bounds__(256) void enf_ode_conv_fwd_kernel(OdeConvArgs A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
  const int ct = blockIdx.x * 4 + wave, r = blockIdx.y, b = blockIdx.z;
  if (ct * 16 >= A.C) return;                                  // wave-uniform, no barrier follows
  const int C = A.C, J = A.J, Z = A.Z;
  float wr[JM][4];                                             // A operand: W[16m + 4 quad + t][16 ct + col]
#pragma unroll
  for (int m = 0; m < JM; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) wr[m][t] = A.W[(size_t)(16 * m + 4 * quad + t) * C + 16 * ct + col];
  const float* kbr = A.kb + (size_t)b * Z * Z * J + (size_t)r * A.sR;
  const float* ab = A.a + (size_t)b * Z * C + 16 * ct + 4 * quad;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < Z; s0 += 16) {
    const int s = s0 + col;
    const bool sv = s < Z;
    const float* kbs = kbr + (size_t)(sv ? s : 0) * A.sS + 4 * quad;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};                            // kernel[c = 16 ct + 4 quad + i][s]
#pragma unroll
    for (int m = 0; m < JM; ++m) {
      f32x4 kv = *reinterpret_cast<const f32x4*>(kbs + 16 * m);
      if (!sv) kv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[m][t], kv[t], d, 0, 0, 0);
    }
    if (sv) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(ab + (size_t)s * C);
      acc += d * av;
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
  }
  if (col == 0) {
    if (A.bias) acc += *reinterpret_cast<const f32x4*>(A.bias + 16 * ct + 4 * quad);
    *reinterpret_cast<f32x4*>(A.out + ((size_t)b * Z + r) * C + 16 * ct + 4 * quad) = acc;
  }
}

// one wave per (b, r, 16-basis tile): d kb[b, r, s, 16 jt + 4 quad + i] for 16 senders at a time
template <int CM>   // C = 16 CM
__global__ __launch_bounds__(256) void enf_ode_conv_dkb_kernel(OdeConvArgs A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
  const int jt = blockIdx.x * 4 + wave, r = blockIdx.y, b = blockIdx.z;
  if (jt * 16 >= A.J) return;
  const int C = A.C, J = A.J, Z = A.Z;
  f32x4 wr[CM], gr[CM];                                        // W[16 jt + col][16 m + 4 quad + t], g[b, r, 16 m + 4 quad + t]
#pragma unroll
  for (int m = 0; m < CM; ++m) {
    wr[m] = *reinterpret_cast<const f32x4*>(A.W + (size_t)(16 * jt + col) * C + 16 * m + 4 * quad);
    gr[m] = *reinterpret_cast<const f32x4*>(A.g + ((size_t)b * Z + r) * C + 16 * m + 4 * quad);
  }
  float* orow = A.out + ((size_t)b * Z + r) * Z * J + 16 * jt + 4 * quad;
  for (int s0 = 0; s0 < Z; s0 += 16) {
    const int s = s0 + col;
    const bool sv = s < Z;
    const float* as = A.a + ((size_t)b * Z + (sv ? s : 0)) * C + 4 * quad;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < CM; ++m) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(as + 16 * m) * gr[m];
#pragma unroll
      for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[m][t], v[t], d, 0, 0, 0);
    }
    if (sv) *reinterpret_cast<f32x4*>(orow + (size_t)s * J) = d;
  }
}

static int ode_check(int B, int Z, int J, int C) {
  if (B <= 0 || Z <= 0) return ENF_EINVAL;
  if (J % 16 || C % 16 || J < 16 || C < 16 || J > 128 || C > 128) return ENF_EUNSUPPORTED;
  if ((J / 16) & (J / 16 - 1)) return ENF_EUNSUPPORTED;      // 16, 32, 64, 128
  if ((C / 16) & (C / 16 - 1)) return ENF_EUNSUPPORTED;
  if (Z > 65535 || B > 65535) return ENF_EUNSUPPORTED;
  return ENF_OK;
}

extern "C" int enf_ode_conv_forward(int B, int Z, int J, int C, const float* a, const float* kb, int64_t kb_stride_r,
                                    int64_t kb_stride_s, const float* W, const float* bias, float* out, void* stream) {
  int rc = ode_check(B, Z, J, C);
  if (rc) return rc;
  if (!a || !kb || !W || !out) return ENF_EINVAL;
  if (kb_stride_r % 4 || kb_stride_s % 4) return ENF_EINVAL;  // 16-byte loads along the basis axis
  OdeConvArgs A{a, kb, W, bias, nullptr, out, B, Z, J, C, (long)kb_stride_r, (long)kb_stride_s};
  const dim3 grid((C / 16 + 3) / 4, Z, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (J / 16) {
    case 1: hipLaunchKernelGGL(enf_ode_conv_fwd_kernel<1>, grid, block, 0, st, A); break;
    case 2: hipLaunchKernelGGL(enf_ode_conv_fwd_kernel<2>, grid, block, 0, st, A); break;
    case 4: hipLaunchKernelGGL(enf_ode_conv_fwd_kernel<4>, grid, block, 0, st, A); break;
    default: hipLaunchKernelGGL(enf_ode_conv_fwd_kernel<8>, grid, block, 0, st, A); break;
  }
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

extern "C" int enf_ode_conv_backward_basis(int B, int Z, int J, int C, const float* a, const float* g, const float* W,
                                           float* dkb, void* stream) {
  int rc = ode_check(B, Z, J, C);
  if (rc) return rc;
  if (!a || !g || !W || !dkb) return ENF_EINVAL;
  OdeConvArgs A{a, nullptr, W, nullptr, g, dkb, B, Z, J, C, 0, 0};
  const dim3 grid((J / 16 + 3) / 4, Z, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (C / 16) {
    case 1: hipLaunchKernelGGL(enf_ode_conv_dkb_kernel<1>, grid, block, 0, st, A); break;
    case 2: hipLaunchKernelGGL(enf_ode_conv_dkb_kernel<2>, grid, block, 0, st, A); break;
    case 4: hipLaunchKernelGGL(enf_ode_conv_dkb_kernel<4>, grid, block, 0, st, A); break;
    default: hipLaunchKernelGGL(enf_ode_conv_dkb_kernel<8>, grid, block, 0, st, A); break;
  }
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}
