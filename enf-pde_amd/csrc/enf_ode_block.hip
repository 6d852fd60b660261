// enf_ode_block.hip -- the per-latent half of a ConvBlock of the latent ODE, fused.
//
// After the separable convolution every ConvBlock of PonitaGen (experiments/fitting/ode_models/ponita_ode_g.py:44-48) runs
//     out = Dense_2(gelu(Dense_1(LayerNorm(x))))          x (R, H), R = B Z latent rows, Dense_1: H -> M = widening * H
// on every latent: 134 MFLOP at the bench shape, run by the op-by-op path as 4 launches forward and 10 backward (library GEMMs,
// LayerNorm, gelu, bias sums), three times per evaluation -- the evaluation is launch-bound.  Here: one kernel forward, one
// kernel + one partial-sum kernel backward (fp32 16x16x4 MFMA, 16 latent rows per 4-wave workgroup):
//   forward : LayerNorm in a row-thread layout -> LDS; pre = xn W1 + b1 (kept for the backward), h = gelu(pre) -> LDS;
//             out = h W2 + b2.  The rows are the MFMA's columns; the waves split the output tiles.
//   backward: LayerNorm again, d h = g W2^T, d pre = d h gelu'(pre), d xn = d pre W1^T, LayerNorm backward -> d x; the weight
//             gradients contract over the 16 rows of the tile (h^T g, xn^T d pre: operands read from the LDS tiles with the rows
//             along K), bias / scale gradients are column sums of the same tiles; one partial per workgroup, summed in a
//             fixed order (enf_ode_sum_partials-style reduction below; bitwise reproducible).
// The sum over the MFMA's K index is order-free: lane quad q takes 4 CONSECUTIVE k of a 16-wide super-step, so operands that
// are contiguous along K are one 16-byte load.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_launch.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define BK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

__device__ __forceinline__ float bk_tanh(float u) {
  const float e = __expf(2.f * u);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float bk_gelu(float x) {      // tanh form (flax nn.gelu default)
  return 0.5f * x * (1.f + bk_tanh(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ void bk_gelu_g(float x, float& y, float& dy) {
  const float x2 = x * x, t = bk_tanh(0.7978845608028654f * (x + 0.044715f * x * x2));
  y = 0.5f * x * (1.f + t);
  dy = 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 0.134145f * x2);
}

struct BkArgs {
  const float* x; const float* gamma; const float* beta;
  const float* W1; const float* b1;            // (H, M), (M)
  const float* W2; const float* b2;            // (M, H), (H)
  float* out;                                  // forward: (R, H)
  float* pre;                                  // forward: writes (R, M); backward: reads it
  const float* g;                              // backward: d out (R, H)
  float* dx;                                   // backward: (R, H)
  float* part;                                 // backward, per workgroup: d W1 (H M) | d W2 (M H) | d b1 (M) | d b2 (H) | d gamma (H) | d beta (H)
  int R; float eps;
};

template <int HT, int MT> struct BkCfg {
  static constexpr int H = 16 * HT, M = 16 * MT, LDH = H + 4, LDM = M + 4;     // padded rows, 16-byte aligned
  static constexpr int PART = 2 * H * M + M + 3 * H;
  static constexpr int LDS_FWD = 4 * 16 * (LDH + LDM);
  static constexpr int LDS_BWD = 4 * 16 * (4 * LDH + 2 * LDM);
};

// LayerNorm of the tile's 16 rows in a row-thread layout: thread (row = tid / 16, part = tid % 16) holds x[row][part + 16 e].
template <int HT>
__device__ __forceinline__ void bk_layer_norm(const BkArgs& A, int row0, int tid, float (&xh)[HT], float& rstd, bool& ok) {
  constexpr int H = 16 * HT;
  const int row = tid >> 4, part = tid & 15;
  ok = row0 + row < A.R;
  const float* xp = A.x + (size_t)(ok ? row0 + row : A.R - 1) * H + part;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < HT; ++e) { xh[e] = xp[16 * e]; s += xh[e]; }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / H);
  float v = 0.f;
#pragma unroll
  for (int e = 0; e < HT; ++e) { xh[e] -= mean; v += xh[e] * xh[e]; }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  rstd = rsqrtf(v * (1.f / H) + A.eps);
#pragma unroll
  for (int e = 0; e < HT; ++e) xh[e] *= rstd;
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int HT, int MT>
__global__ __launch_bounds__(256) void enf_ode_block_fwd_kernel(BkArgs A) {
  using C = BkCfg<HT, MT>;
  constexpr int H = C::H, M = C::M, LDH = C::LDH, LDM = C::LDM;
  __shared__ __attribute__((aligned(16))) float sXn[16 * LDH];
  __shared__ __attribute__((aligned(16))) float sH[16 * LDM];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * 16;
  {
    float xh[HT], rstd;
    bool ok;
    bk_layer_norm<HT>(A, row0, tid, xh, rstd, ok);
    const int row = tid >> 4, part = tid & 15;
#pragma unroll
    for (int e = 0; e < HT; ++e) sXn[row * LDH + part + 16 * e] = xh[e] * A.gamma[part + 16 * e] + A.beta[part + 16 * e];
  }
  __syncthreads();
  const bool rok = row0 + col < A.R;
  // pre^T tiles (16 hidden units x 16 rows): A = W1 (k = input feature), B = xn^T
#pragma unroll
  for (int i = 0; i < (MT + 3) / 4; ++i) {
    const int tau = wave + 4 * i;
    if (tau < MT) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(A.b1 + 16 * tau + 4 * quad);
      for (int s = 0; s < HT; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sXn + col * LDH + 16 * s + 4 * quad);
        const float* w = A.W1 + (size_t)(16 * s + 4 * quad) * M + 16 * tau + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = BK_MFMA(w[(size_t)j * M], b[j], acc);
      }
      if (rok) *reinterpret_cast<f32x4*>(A.pre + (size_t)(row0 + col) * M + 16 * tau + 4 * quad) = acc;
      f32x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = bk_gelu(acc[r]);
      *reinterpret_cast<f32x4*>(sH + col * LDM + 16 * tau + 4 * quad) = h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < (HT + 3) / 4; ++i) {
    const int tau = wave + 4 * i;
    if (tau < HT) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(A.b2 + 16 * tau + 4 * quad);
      for (int s = 0; s < MT; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sH + col * LDM + 16 * s + 4 * quad);
        const float* w = A.W2 + (size_t)(16 * s + 4 * quad) * H + 16 * tau + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = BK_MFMA(w[(size_t)j * H], b[j], acc);
      }
      if (rok) *reinterpret_cast<f32x4*>(A.out + (size_t)(row0 + col) * H + 16 * tau + 4 * quad) = acc;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
template <int HT, int MT>
__global__ __launch_bounds__(256) void enf_ode_block_bwd_kernel(BkArgs A) {
  using C = BkCfg<HT, MT>;
  constexpr int H = C::H, M = C::M, LDH = C::LDH, LDM = C::LDM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sXn = reinterpret_cast<float*>(smem);       // LayerNorm output
  float* sXh = sXn + 16 * LDH;                        // normalised x (before scale / bias)
  float* sG = sXh + 16 * LDH;                         // d out
  float* sDy = sG + 16 * LDH;                         // d (LayerNorm output)
  float* sH = sDy + 16 * LDH;                         // gelu(pre)
  float* sDp = sH + 16 * LDM;                         // d pre
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * 16, row = tid >> 4, part = tid & 15;
  float xh[HT], rstd;
  bool ok;
  bk_layer_norm<HT>(A, row0, tid, xh, rstd, ok);
#pragma unroll
  for (int e = 0; e < HT; ++e) {
    const int h = part + 16 * e;
    sXh[row * LDH + h] = xh[e];
    sXn[row * LDH + h] = xh[e] * A.gamma[h] + A.beta[h];
    sG[row * LDH + h] = ok ? A.g[(size_t)(row0 + row) * H + h] : 0.f;       // rows past R contribute nothing anywhere
  }
  __syncthreads();
  const bool rok = row0 + col < A.R;
  // d h^T tiles: A = W2 (k = output feature, contiguous), B = g^T;  d pre = d h gelu'(pre), h = gelu(pre)
#pragma unroll
  for (int i = 0; i < (MT + 3) / 4; ++i) {
    const int tau = wave + 4 * i;
    if (tau < MT) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < HT; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sG + col * LDH + 16 * s + 4 * quad);
        const f32x4 w = *reinterpret_cast<const f32x4*>(A.W2 + (size_t)(16 * tau + col) * H + 16 * s + 4 * quad);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = BK_MFMA(w[j], b[j], acc);
      }
      const f32x4 pre = *reinterpret_cast<const f32x4*>(A.pre + (size_t)(rok ? row0 + col : A.R - 1) * M + 16 * tau + 4 * quad);
      f32x4 h, dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y, dy;
        bk_gelu_g(pre[r], y, dy);
        h[r] = y;
        dp[r] = acc[r] * dy;
      }
      *reinterpret_cast<f32x4*>(sH + col * LDM + 16 * tau + 4 * quad) = h;
      *reinterpret_cast<f32x4*>(sDp + col * LDM + 16 * tau + 4 * quad) = dp;
    }
  }
  __syncthreads();
  // d xn^T tiles: A = W1 (k = hidden unit, contiguous), B = d pre^T
#pragma unroll
  for (int i = 0; i < (HT + 3) / 4; ++i) {
    const int tau = wave + 4 * i;
    if (tau < HT) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < MT; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sDp + col * LDM + 16 * s + 4 * quad);
        const f32x4 w = *reinterpret_cast<const f32x4*>(A.W1 + (size_t)(16 * tau + col) * M + 16 * s + 4 * quad);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = BK_MFMA(w[j], b[j], acc);
      }
      *reinterpret_cast<f32x4*>(sDy + col * LDH + 16 * tau + 4 * quad) = acc;
    }
  }
  __syncthreads();
  // LayerNorm backward (row-thread layout again; xh[] still holds this thread's normalised values)
  {
    float dxh[HT], c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int e = 0; e < HT; ++e) {
      dxh[e] = sDy[row * LDH + part + 16 * e] * A.gamma[part + 16 * e];
      c1 += dxh[e];
      c2 += dxh[e] * xh[e];
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    c1 *= 1.f / H;
    c2 *= 1.f / H;
    if (ok) {
#pragma unroll
      for (int e = 0; e < HT; ++e) A.dx[(size_t)(row0 + row) * H + part + 16 * e] = rstd * (dxh[e] - c1 - xh[e] * c2);
    }
  }
  float* part1 = A.part + (size_t)blockIdx.x * C::PART;
  float* part2 = part1 + H * M;
  float* pvec = part2 + M * H;
  // column sums over the 16 rows: d b1 | d b2 | d gamma | d beta
  for (int c = tid; c < M + 3 * H; c += 256) {
    float s = 0.f;
    if (c < M) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s += sDp[r * LDM + c];
    } else if (c < M + H) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s += sG[r * LDH + c - M];
    } else if (c < M + 2 * H) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s += sDy[r * LDH + c - M - H] * sXh[r * LDH + c - M - H];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s += sDy[r * LDH + c - M - 2 * H];
    }
    pvec[c] = s;
  }
  // d W1[h][m] = sum_r xn[r][h] d pre[r][m];  d W2[m][h] = sum_r h[r][m] g[r][h]: the rows are K (lane quad q: rows 4 q + j)
  for (int i = 0; i < (HT * MT + 3) / 4; ++i) {
    const int tau = wave + 4 * i;
    if (tau < HT * MT) {
      const int ht = tau % HT, mt = tau / HT;
      f32x4 a1 = f32x4{0.f, 0.f, 0.f, 0.f}, a2 = a1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * quad + j;
        a1 = BK_MFMA(sXn[r * LDH + 16 * ht + col], sDp[r * LDM + 16 * mt + col], a1);
        a2 = BK_MFMA(sH[r * LDM + 16 * mt + col], sG[r * LDH + 16 * ht + col], a2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        part1[(size_t)(16 * ht + 4 * quad + r) * M + 16 * mt + col] = a1[r];
        part2[(size_t)(16 * mt + 4 * quad + r) * H + 16 * ht + col] = a2[r];
      }
    }
  }
}

// out[e] = sum over w of part[w][e]: 32 elements x 8 groups per block, fixed order
__global__ __launch_bounds__(256) void enf_ode_block_sum_kernel(const float* part, int nwg, int n, float* out, int accumulate) {
  __shared__ float red[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  float s = 0.f;
  if (e < n)
    for (int w = grp; w < nwg; w += 8) s += part[(size_t)w * n + e];
  red[grp][threadIdx.x & 31] = s;
  __syncthreads();
  if (grp == 0 && e < n) {
    float t = 0.f;
    for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x & 31];
    out[e] = accumulate ? out[e] + t : t;
  }
}

// --------------------------------------------------------------------------------------------------------------- host
constexpr int BK_CHUNK_WGS = 256;       // rows per backward chunk = 16 x this

extern "C" int enf_ode_block_supported(int H, int M) {
  return (H == 32 || H == 64 || H == 128) && M == 2 * H ? 1 : 0;
}
extern "C" size_t enf_ode_block_scratch_bytes(int64_t R, int H, int M) {
  if (!enf_ode_block_supported(H, M) || R <= 0) return 0;
  // one partial (2 H M + M + 3 H floats, 265 KB at H = 128) per 16-row workgroup of a CHUNK of at most BK_CHUNK_WGS workgroups:
  // the backward walks larger inputs chunk by chunk, so the scratch stays bounded (68 MB at H = 128) whatever R is
  const int64_t nwg = (R + 15) / 16;
  return 4 * (size_t)(nwg < BK_CHUNK_WGS ? nwg : BK_CHUNK_WGS) * (2 * (size_t)H * M + M + 3 * H);
}

extern "C" int enf_ode_block_forward(int64_t R, int H, int M, const float* x, const float* gamma, const float* beta,
                                     const float* W1, const float* b1, const float* W2, const float* b2, float eps, float* out,
                                     float* pre, void* stream) {
  if (R <= 0 || R > (1ll << 30)) return ENF_EINVAL;
  if (!enf_ode_block_supported(H, M)) return ENF_EUNSUPPORTED;
  if (!x || !gamma || !beta || !W1 || !b1 || !W2 || !b2 || !out || !pre) return ENF_EINVAL;
  BkArgs A{x, gamma, beta, W1, b1, W2, b2, out, pre, nullptr, nullptr, nullptr, (int)R, eps};
  const dim3 grid((unsigned)((R + 15) / 16)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (H == 32) hipLaunchKernelGGL((enf_ode_block_fwd_kernel<2, 4>), grid, block, 0, st, A);
  else if (H == 64) hipLaunchKernelGGL((enf_ode_block_fwd_kernel<4, 8>), grid, block, 0, st, A);
  else hipLaunchKernelGGL((enf_ode_block_fwd_kernel<8, 16>), grid, block, 0, st, A);
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

template <int HT, int MT> static int bk_launch_bwd(const BkArgs& A, dim3 grid, hipStream_t st) {
  static EnfAttrBits attr{0};
  constexpr int LDS = BkCfg<HT, MT>::LDS_BWD;
  if (!enf_lds_attr((const void*)enf_ode_block_bwd_kernel<HT, MT>, LDS, attr)) return ENF_ELAUNCH;
  hipLaunchKernelGGL((enf_ode_block_bwd_kernel<HT, MT>), grid, dim3(256), LDS, st, A);
  return ENF_OK;
}

// dparams: one buffer of 2 H M + M + 3 H floats: d W1 (H, M) | d W2 (M, H) | d b1 | d b2 | d gamma | d beta
extern "C" int enf_ode_block_backward(int64_t R, int H, int M, const float* x, const float* gamma, const float* beta,
                                      const float* W1, const float* W2, const float* pre, const float* g, float eps, float* dx,
                                      float* dparams, void* scratch, size_t scratch_bytes, void* stream) {
  if (R <= 0 || R > (1ll << 30)) return ENF_EINVAL;
  if (!enf_ode_block_supported(H, M)) return ENF_EUNSUPPORTED;
  if (!x || !gamma || !beta || !W1 || !W2 || !pre || !g || !dx || !dparams || !scratch) return ENF_EINVAL;
  if (scratch_bytes < enf_ode_block_scratch_bytes(R, H, M) || ((uintptr_t)scratch & 15)) return ENF_EINVAL;
  const int n = 2 * H * M + M + 3 * H;
  hipStream_t st = (hipStream_t)stream;
  const int64_t chunk_rows = 16ll * BK_CHUNK_WGS;
  for (int64_t r0 = 0; r0 < R; r0 += chunk_rows) {
    const int rows = (int)(R - r0 < chunk_rows ? R - r0 : chunk_rows);
    BkArgs A{x + r0 * H, gamma, beta, W1, nullptr, W2, nullptr, nullptr, const_cast<float*>(pre) + r0 * M, g + r0 * H, dx + r0 * H,
             (float*)scratch, rows, eps};
    const int nwg = (rows + 15) / 16;
    int rc;
    if (H == 32) rc = bk_launch_bwd<2, 4>(A, dim3(nwg), st);
    else if (H == 64) rc = bk_launch_bwd<4, 8>(A, dim3(nwg), st);
    else rc = bk_launch_bwd<8, 16>(A, dim3(nwg), st);
    if (rc) return rc;
    // (stream order: the next chunk's partials overwrite the scratch only after this sum has read it)
    hipLaunchKernelGGL(enf_ode_block_sum_kernel, dim3((n + 31) / 32), dim3(256), 0, st, (const float*)scratch, nwg, n, dparams, r0 > 0 ? 1 : 0);
  }
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}
