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
//   backward, weight : d W = kb^T (g (x) a) over the pair axis, the right operand formed in registers (enf_ode_conv_dw_kernel).
// The sum over the MFMA's K index is order-free, so each lane's four K-steps use four
// CONSECUTIVE basis functions / channels (one 16-byte load feeds four MFMAs).
// (First version: one wave per (b, r, channel tile) -- 8 re-reads of every basis tile and one dependent MFMA chain per
// wave: 22.5 us per call at B=16, Z=64, J=64, C=128 = 0.31 of the fp32-MFMA peak.)
#include <hip/hip_runtime.h>
#include "enf_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define OB_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

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

// One wave per (b, r) and group of CG 16-channel tiles; the 4 waves of a block take 4 receivers.  A sender tile's basis
// values are loaded once and meet CG independent accumulators (the 16x16x4 f32 MFMA has a 40-cycle dependent latency
// against a 32-cycle issue: independent chains keep the pipe full), W stays in registers for the whole sweep.
template <int JM, int CG>   // J = 16 JM; CG channel tiles per wave (CG * JM * 4 <= 128 registers of W)
__global__ __launch_bounds__(256) void enf_ode_conv_fwd_kernel(OdeConvArgs A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
  const int ct0 = blockIdx.x * CG, r = blockIdx.y * 4 + wave, b = blockIdx.z;
  const int C = A.C, J = A.J, Z = A.Z;
  if (r >= Z) return;                                          // wave-uniform, no barrier follows
  float wr[CG][JM][4];                                         // A operand: W[16m + 4 quad + t][16 ct + col]
#pragma unroll
  for (int g = 0; g < CG; ++g)
#pragma unroll
    for (int m = 0; m < JM; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        wr[g][m][t] = 16 * (ct0 + g) < C ? A.W[(size_t)(16 * m + 4 * quad + t) * C + 16 * (ct0 + g) + col] : 0.f;
  const float* kbr = A.kb + (size_t)b * Z * Z * J + (size_t)r * A.sR;
  const float* ab = A.a + (size_t)b * Z * C + 4 * quad;
  f32x4 acc[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < Z; s0 += 16) {
    const int s = s0 + col;
    const bool sv = s < Z;
    const float* kbs = kbr + (size_t)(sv ? s : 0) * A.sS + 4 * quad;
    f32x4 kv[JM];
#pragma unroll
    for (int m = 0; m < JM; ++m) {
      kv[m] = *reinterpret_cast<const f32x4*>(kbs + 16 * m);
      if (!sv) kv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 d[CG];                                               // kernel[c = 16 ct + 4 quad + i][s]
#pragma unroll
    for (int g = 0; g < CG; ++g) d[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < JM; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < CG; ++g) d[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][m][t], kv[m][t], d[g], 0, 0, 0);
    if (sv) {
#pragma unroll
      for (int g = 0; g < CG; ++g)
        if (16 * (ct0 + g) < C) acc[g] += d[g] * *reinterpret_cast<const f32x4*>(ab + (size_t)s * C + 16 * (ct0 + g));
    }
  }
#pragma unroll
  for (int g = 0; g < CG; ++g) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[g][i] += __shfl_xor(acc[g][i], o, 64);
    }
    if (col == 0 && 16 * (ct0 + g) < C) {
      if (A.bias) acc[g] += *reinterpret_cast<const f32x4*>(A.bias + 16 * (ct0 + g) + 4 * quad);
      *reinterpret_cast<f32x4*>(A.out + ((size_t)b * Z + r) * C + 16 * (ct0 + g) + 4 * quad) = acc[g];
    }
  }
}

// One wave per (b, r) and group of JG 16-basis tiles: d kb[b, r, s, 16 jt + 4 quad + i] for 16 senders at a time; the
// product g (.) a of a sender tile is formed once and meets JG independent accumulators.
template <int CM, int JG>   // C = 16 CM; JG basis tiles per wave (JG * CM * 4 <= 128 registers of W)
__global__ __launch_bounds__(256) void enf_ode_conv_dkb_kernel(OdeConvArgs A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
  const int jt0 = blockIdx.x * JG, r = blockIdx.y * 4 + wave, b = blockIdx.z;
  const int C = A.C, J = A.J, Z = A.Z;
  if (r >= Z) return;
  f32x4 wr[JG][CM], gr[CM];                                    // W[16 jt + col][16 m + 4 quad + t], g[b, r, 16 m + 4 quad + t]
#pragma unroll
  for (int m = 0; m < CM; ++m) {
    gr[m] = *reinterpret_cast<const f32x4*>(A.g + ((size_t)b * Z + r) * C + 16 * m + 4 * quad);
#pragma unroll
    for (int g = 0; g < JG; ++g)
      wr[g][m] = 16 * (jt0 + g) < J ? *reinterpret_cast<const f32x4*>(A.W + (size_t)(16 * (jt0 + g) + col) * C + 16 * m + 4 * quad)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float* orow = A.out + ((size_t)b * Z + r) * Z * J + 4 * quad;
  for (int s0 = 0; s0 < Z; s0 += 16) {
    const int s = s0 + col;
    const bool sv = s < Z;
    const float* as = A.a + ((size_t)b * Z + (sv ? s : 0)) * C + 4 * quad;
    f32x4 d[JG];
#pragma unroll
    for (int g = 0; g < JG; ++g) d[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < CM; ++m) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(as + 16 * m) * gr[m];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < JG; ++g) d[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[g][m][t], v[t], d[g], 0, 0, 0);
    }
    if (sv) {
#pragma unroll
      for (int g = 0; g < JG; ++g)
        if (16 * (jt0 + g) < J) *reinterpret_cast<f32x4*>(orow + (size_t)s * J + 16 * (jt0 + g)) = d[g];
    }
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
  // channel tiles per wave: all of them while W fits 128 registers, fewer (more waves) when B Z alone cannot fill the chip
  const int CT = C / 16, JM = J / 16;
  int cg = CT;
  while (cg * JM * 4 > 128) cg /= 2;
  while (cg > 1 && (long)B * Z * (CT / cg) < 2048) cg /= 2;
  const dim3 grid(CT / cg, (Z + 3) / 4, B), block(256);
  hipStream_t st = (hipStream_t)stream;
#define ODE_FWD(JM_, CG_) if (JM == JM_ && cg == CG_) hipLaunchKernelGGL((enf_ode_conv_fwd_kernel<JM_, CG_>), grid, block, 0, st, A);
  ODE_FWD(1, 1) ODE_FWD(1, 2) ODE_FWD(1, 4) ODE_FWD(1, 8)
  ODE_FWD(2, 1) ODE_FWD(2, 2) ODE_FWD(2, 4) ODE_FWD(2, 8)
  ODE_FWD(4, 1) ODE_FWD(4, 2) ODE_FWD(4, 4) ODE_FWD(4, 8)
  ODE_FWD(8, 1) ODE_FWD(8, 2) ODE_FWD(8, 4)
#undef ODE_FWD
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

extern "C" int enf_ode_conv_backward_basis(int B, int Z, int J, int C, const float* a, const float* g, const float* W,
                                           float* dkb, void* stream) {
  int rc = ode_check(B, Z, J, C);
  if (rc) return rc;
  if (!a || !g || !W || !dkb) return ENF_EINVAL;
  OdeConvArgs A{a, nullptr, W, nullptr, g, dkb, B, Z, J, C, 0, 0};
  const int CM = C / 16, JT = J / 16;
  int jg = JT;
  while (jg * CM * 4 > 128) jg /= 2;
  while (jg > 1 && (long)B * Z * (JT / jg) < 2048) jg /= 2;
  const dim3 grid(JT / jg, (Z + 3) / 4, B), block(256);
  hipStream_t st = (hipStream_t)stream;
#define ODE_DKB(CM_, JG_) if (CM == CM_ && jg == JG_) hipLaunchKernelGGL((enf_ode_conv_dkb_kernel<CM_, JG_>), grid, block, 0, st, A);
  ODE_DKB(1, 1) ODE_DKB(1, 2) ODE_DKB(1, 4) ODE_DKB(1, 8)
  ODE_DKB(2, 1) ODE_DKB(2, 2) ODE_DKB(2, 4) ODE_DKB(2, 8)
  ODE_DKB(4, 1) ODE_DKB(4, 2) ODE_DKB(4, 4) ODE_DKB(4, 8)
  ODE_DKB(8, 1) ODE_DKB(8, 2) ODE_DKB(8, 4)
#undef ODE_DKB
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

// d W[j][c] = sum over (b, r, s) of kb[b,r,s,j] * g[b,r,c] * a[b,s,c]  (SepGconv's weight gradient): a (J x P)(P x C) product
// over the P = B Z^2 pairs whose right operand g (x) a is formed in registers (the unfused path writes it out: 33 MB per layer
// at the bench shape, then a split-K library GEMM and two reductions).  The pairs are the MFMA's K index: lane quad q takes
// senders s0 + 4 q + 0..3 of a 16-sender tile.  A workgroup walks a contiguous share of the (b, r) rows; wave w owns the
// channel tiles w, w + 4 for every basis tile, accumulates them in registers over the whole share and writes one partial
// (J, C) per workgroup -- followed by its share of d bias[c] = sum_{b,r} g[b,r,c], which costs the kernel one add per row --;
// enf_ode_sum_partials_kernel adds the partials in a fixed order.
struct OdeConvDwArgs { const float* a; const float* kb; const float* g; float* part; int B, Z, J, C, rows_per_wg; };

template <int JT, int CK>   // J = 16 JT; CK = channel tiles per wave (C <= 64 CK)
__global__ __launch_bounds__(256) void enf_ode_conv_dw_kernel(OdeConvDwArgs A) {
  const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = A.C, J = A.J, Z = A.Z, CT = C / 16;
  f32x4 acc[CK][JT];
#pragma unroll
  for (int k = 0; k < CK; ++k)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) acc[k][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int row0 = blockIdx.x * A.rows_per_wg, row1 = min(row0 + A.rows_per_wg, A.B * Z);
  const int ST = (Z + 15) / 16, nstep = (row1 - row0) * ST;        // steps = (row, 16-sender tile); operands fetched one step ahead
  f32x4 av[JT], aa[CK], avn[JT], aan[CK];
  float gv[CK], gvn[CK], dbias[CK];                              // d bias[c] = sum over the rows of g[row][c]: rides along
#pragma unroll
  for (int k = 0; k < CK; ++k) dbias[k] = 0.f;
  auto fetch = [&](int step, f32x4 (&va)[JT], f32x4 (&vb)[CK], float (&vg)[CK]) {
    const int row = row0 + step / ST, s0 = 16 * (step % ST), b = row / Z;
    const float* kbr = A.kb + (size_t)row * Z * J;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = s0 + 4 * quad + r, sc = s < Z ? s : Z - 1;
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) va[jt][r] = kbr[(size_t)sc * J + 16 * jt + col];
#pragma unroll
      for (int k = 0; k < CK; ++k)
        vb[k][r] = (s < Z && wave + 4 * k < CT) ? A.a[((size_t)b * Z + s) * C + 16 * (wave + 4 * k) + col] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < CK; ++k) vg[k] = wave + 4 * k < CT ? A.g[(size_t)row * C + 16 * (wave + 4 * k) + col] : 0.f;
  };
  if (nstep > 0) fetch(0, av, aa, gv);
  for (int step = 0; step < nstep; ++step) {
    fetch(step + 1 < nstep ? step + 1 : step, avn, aan, gvn);
    if (step % ST == 0) {                                        // first sender tile of a row
#pragma unroll
      for (int k = 0; k < CK; ++k) dbias[k] += gv[k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int k = 0; k < CK; ++k) acc[k][jt] = OB_MFMA4(av[jt][r], gv[k] * aa[k][r], acc[k][jt]);
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) av[jt] = avn[jt];
#pragma unroll
    for (int k = 0; k < CK; ++k) { aa[k] = aan[k]; gv[k] = gvn[k]; }
  }
  float* part = A.part + (size_t)blockIdx.x * (J * C + C);
#pragma unroll
  for (int k = 0; k < CK; ++k) {
    if (wave + 4 * k < CT) {
      if (quad == 0) part[(size_t)J * C + 16 * (wave + 4 * k) + col] = dbias[k];
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(size_t)(16 * jt + 4 * quad + r) * C + 16 * (wave + 4 * k) + col] = acc[k][jt][r];
    }
  }
}

// out[e] = sum over w of part[w][e], e < n: 32 elements x 8 groups per block, each group a fixed stride-8 share, then the 8
// group sums in order (bitwise reproducible).
__global__ __launch_bounds__(256) void enf_ode_sum_partials_kernel(const float* part, int nwg, int n, float* out) {
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
    out[e] = t;
  }
}

static int ode_dw_wgs(int B, int Z) { return B * Z < 256 ? B * Z : 256; }

extern "C" size_t enf_ode_conv_backward_weight_scratch_bytes(int B, int Z, int J, int C) {
  if (ode_check(B, Z, J, C)) return 0;
  return (size_t)ode_dw_wgs(B, Z) * (J * C + C) * 4;
}

extern "C" int enf_ode_conv_backward_weight(int B, int Z, int J, int C, const float* a, const float* kb, const float* g,
                                            float* dW, void* scratch, size_t scratch_bytes, void* stream) {
  int rc = ode_check(B, Z, J, C);
  if (rc) return rc;
  if (!a || !kb || !g || !dW || !scratch) return ENF_EINVAL;
  if (scratch_bytes < enf_ode_conv_backward_weight_scratch_bytes(B, Z, J, C)) return ENF_EINVAL;
  const int rows = B * Z, nwg0 = ode_dw_wgs(B, Z), rpw = (rows + nwg0 - 1) / nwg0, nwg = (rows + rpw - 1) / rpw;
  OdeConvDwArgs A{a, kb, g, (float*)scratch, B, Z, J, C, rpw};
  const int JT = J / 16, CK = C > 64 ? 2 : 1;
  hipStream_t st = (hipStream_t)stream;
#define ODE_DW(JT_, CK_) if (JT == JT_ && CK == CK_) hipLaunchKernelGGL((enf_ode_conv_dw_kernel<JT_, CK_>), dim3(nwg), dim3(256), 0, st, A);
  ODE_DW(1, 1) ODE_DW(2, 1) ODE_DW(4, 1) ODE_DW(8, 1) ODE_DW(1, 2) ODE_DW(2, 2) ODE_DW(4, 2) ODE_DW(8, 2)
#undef ODE_DW
  hipLaunchKernelGGL(enf_ode_sum_partials_kernel, dim3((J * C + C + 31) / 32), dim3(256), 0, st, (const float*)scratch, nwg, J * C + C, dW);
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Polynomial features of the pair invariants (PolynomialFeatures, ponita_ode_g.py:15-26): the Kronecker powers
// [x, x(x)x, (x(x)x)(x)x, ...] of an I-vector, degree + 1 blocks, F = I + I^2 + ... values per pair.  Feature k of block
// d (0-based) is the product of d + 1 components whose indices are the base-I digits of its offset in the block, most
// significant first (the reference's einsum('...i,...j->...ij', prev, x) appends the new factor as the LAST index).
// One 64-lane wave per pair, lanes stride over the features (coalesced).  Backward: d x_i accumulates
// dF_k * prod_{other factors} for every position holding index i (prefix / suffix products), then a wave reduction.  The reference materialises every intermediate power (and autograd their gradients).
struct OdePolyArgs { const float* x; const float* dF; float* out; long P; int I, degree, F; };

// I is a template parameter: the digit arithmetic divides by a constant, and component arrays are indexed statically
// (a runtime I put them in scratch memory: 381 us for the backward at 65,536 pairs x 340 features)
template <int I> __device__ __forceinline__ void ode_poly_decode(int k, int& d, int& off) {
  d = 0;
  int blk = I;
  off = k;
  while (off >= blk) { off -= blk; blk *= I; ++d; }
}
template <int I> __device__ __forceinline__ float ode_poly_pick(const float (&x)[I], int i) {
  float v = x[0];
#pragma unroll
  for (int c = 1; c < I; ++c) v = i == c ? x[c] : v;
  return v;
}

template <int I>
__global__ __launch_bounds__(256) void enf_ode_poly_fwd_kernel(OdePolyArgs A) {
  // one wave per pair, lanes stride over the features (coalesced stores; no per-element 64-bit division)
  const int lane = threadIdx.x & 63;
  const long p = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= A.P) return;                                              // wave-uniform
  float x[I];
#pragma unroll
  for (int c = 0; c < I; ++c) x[c] = A.x[p * I + c];
  float* out = A.out + p * A.F;
  for (int k = lane; k < A.F; k += 64) {
    int d, off;
    ode_poly_decode<I>(k, d, off);
    float v = 1.f;
    for (int j = 0; j <= d; ++j) { v *= ode_poly_pick<I>(x, off % I); off /= I; }   // (the order of the factors is irrelevant)
    out[k] = v;
  }
}

template <int I>
__global__ __launch_bounds__(256) void enf_ode_poly_bwd_kernel(OdePolyArgs A) {
  const int lane = threadIdx.x & 63;
  const long p = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= A.P) return;                                              // wave-uniform
  float x[I], g[I];
#pragma unroll
  for (int c = 0; c < I; ++c) { x[c] = A.x[p * I + c]; g[c] = 0.f; }
  const float* dF = A.dF + p * A.F;
  for (int k = lane; k < A.F; k += 64) {
    int d, off;
    ode_poly_decode<I>(k, d, off);
    const float gk = dF[k];
    // prefix / suffix products over the factor positions: the derivative w.r.t. position j is gk * pre[j] * suf[j]
    int dig[8];
    float pre[8];
    float run = gk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j <= d) { dig[j] = off % I; off /= I; pre[j] = run; run *= ode_poly_pick<I>(x, dig[j]); }
    }
    float suf = 1.f;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
      if (j <= d) {
        const float v = pre[j] * suf;
#pragma unroll
        for (int c = 0; c < I; ++c) g[c] += dig[j] == c ? v : 0.f;
        suf *= ode_poly_pick<I>(x, dig[j]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < I; ++c) {
    float v = g[c];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) A.out[p * I + c] = v;
  }
}

extern "C" int enf_ode_poly_num_features(int I, int degree) {
  if (I < 1 || I > 8 || degree < 0 || degree > 7) return ENF_EUNSUPPORTED;
  long f = 0, b = 1;
  for (int d = 0; d <= degree; ++d) { b *= I; f += b; if (f > (1 << 20)) return ENF_EUNSUPPORTED; }
  return (int)f;
}

extern "C" int enf_ode_poly_forward(int64_t P, int I, int degree, const float* x, float* feat, void* stream) {
  const int F = enf_ode_poly_num_features(I, degree);
  if (F < 0) return F;
  if (P <= 0 || !x || !feat) return ENF_EINVAL;
  OdePolyArgs A{x, nullptr, feat, (long)P, I, degree, F};
  const dim3 grid((unsigned)((P + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (I) {
    case 1: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<1>, grid, block, 0, st, A); break;
    case 2: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<2>, grid, block, 0, st, A); break;
    case 3: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<3>, grid, block, 0, st, A); break;
    case 4: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<4>, grid, block, 0, st, A); break;
    case 5: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<5>, grid, block, 0, st, A); break;
    case 6: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<6>, grid, block, 0, st, A); break;
    case 7: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<7>, grid, block, 0, st, A); break;
    default: hipLaunchKernelGGL(enf_ode_poly_fwd_kernel<8>, grid, block, 0, st, A); break;
  }
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

extern "C" int enf_ode_poly_backward(int64_t P, int I, int degree, const float* x, const float* dfeat, float* dx, void* stream) {
  const int F = enf_ode_poly_num_features(I, degree);
  if (F < 0) return F;
  if (P <= 0 || !x || !dfeat || !dx) return ENF_EINVAL;
  OdePolyArgs A{x, dfeat, dx, (long)P, I, degree, F};
  const dim3 grid((unsigned)((P + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (I) {
    case 1: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<1>, grid, block, 0, st, A); break;
    case 2: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<2>, grid, block, 0, st, A); break;
    case 3: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<3>, grid, block, 0, st, A); break;
    case 4: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<4>, grid, block, 0, st, A); break;
    case 5: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<5>, grid, block, 0, st, A); break;
    case 6: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<6>, grid, block, 0, st, A); break;
    case 7: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<7>, grid, block, 0, st, A); break;
    default: hipLaunchKernelGGL(enf_ode_poly_bwd_kernel<8>, grid, block, 0, st, A); break;
  }
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Vector readout of PonitaGen (ponita_ode_g.py:176-193):  out[b,r,:] = mean_s  wgt[b,r,s] * v[b,r,s,:]  with the pair weight
//     wgt[b,r,s] = inv[b,r,s,:] . Wi + aw[b,s]        (Dense([invariants | a_s]) with one output; aw = a @ W[I:] per latent)
// and the pair vector v = cr * u[b,r,:] + cs * w[b,s,:]  (relative position: u = w = p, cr = 1, cs = -1; sender orientation:
// cr = 0, cs = 1).  The op-by-op path spends ~10 launches forward and ~25 backward on (B, Z, Z, .) intermediates; here one
// launch each way, no atomics: a workgroup owns 64 receivers (4 threads each, senders strided) and, in the backward, the
// matching 64 senders for the sums over receivers.  d Wi comes out as one partial per workgroup.
struct OdeVecArgs {
  const float* inv; const float* aw; const float* u; const float* w; const float* Wi; const float* g;
  float* out; float* dinv; float* daw; float* du; float* dw; float* dWi_part;
  int B, Z, I, D; float cr, cs;
};

template <int I, int D>
__global__ __launch_bounds__(256) void enf_ode_vec_fwd_kernel(OdeVecArgs A) {
  const int b = blockIdx.x, r = blockIdx.y * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3, Z = A.Z;
  const bool ok = r < Z;
  const int rc = ok ? r : Z - 1;
  float wi[I], ur[D], acc[D];
#pragma unroll
  for (int i = 0; i < I; ++i) wi[i] = A.Wi[i];
#pragma unroll
  for (int d = 0; d < D; ++d) { ur[d] = A.cr * A.u[((size_t)b * Z + rc) * D + d]; acc[d] = 0.f; }
  const float* ip = A.inv + ((size_t)b * Z + rc) * Z * I;
  for (int s = part; s < Z; s += 4) {
    float wgt = A.aw[(size_t)b * Z + s];
#pragma unroll
    for (int i = 0; i < I; ++i) wgt += ip[(size_t)s * I + i] * wi[i];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] += wgt * (ur[d] + A.cs * A.w[((size_t)b * Z + s) * D + d]);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    acc[d] += __shfl_xor(acc[d], 1, 64);
    acc[d] += __shfl_xor(acc[d], 2, 64);
    if (ok && part == 0) A.out[((size_t)b * Z + r) * D + d] = acc[d] / Z;
  }
}

template <int I, int D>
__global__ __launch_bounds__(256) void enf_ode_vec_bwd_kernel(OdeVecArgs A) {
  __shared__ float red[64][I];
  const int b = blockIdx.x, t = threadIdx.x >> 2, part = threadIdx.x & 3, Z = A.Z;
  const int q = blockIdx.y * 64 + t;                    // receiver index in pass 1, sender index in pass 2
  const bool ok = q < Z;
  const int qc = ok ? q : Z - 1;
  const float invZ = 1.f / Z;
  float wi[I];
#pragma unroll
  for (int i = 0; i < I; ++i) wi[i] = A.Wi[i];
  // ---- pass 1, receiver q: d inv, d u, d Wi
  {
    float gr[D], ur[D], dwi[I], sw = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { gr[d] = ok ? A.g[((size_t)b * Z + q) * D + d] * invZ : 0.f; ur[d] = A.cr * A.u[((size_t)b * Z + qc) * D + d]; }
#pragma unroll
    for (int i = 0; i < I; ++i) dwi[i] = 0.f;
    const size_t row = ((size_t)b * Z + qc) * Z;
    for (int s = part; s < Z; s += 4) {
      float x[I], wgt = A.aw[(size_t)b * Z + s], dwgt = 0.f;
#pragma unroll
      for (int i = 0; i < I; ++i) { x[i] = A.inv[(row + s) * I + i]; wgt += x[i] * wi[i]; }
#pragma unroll
      for (int d = 0; d < D; ++d) dwgt += gr[d] * (ur[d] + A.cs * A.w[((size_t)b * Z + s) * D + d]);
      sw += wgt;
#pragma unroll
      for (int i = 0; i < I; ++i) {
        dwi[i] += dwgt * x[i];
        if (ok) A.dinv[(row + s) * I + i] = dwgt * wi[i];
      }
    }
    sw += __shfl_xor(sw, 1, 64);
    sw += __shfl_xor(sw, 2, 64);
    if (ok && part == 0) {
#pragma unroll
      for (int d = 0; d < D; ++d) A.du[((size_t)b * Z + q) * D + d] = A.cr * gr[d] * sw;
    }
#pragma unroll
    for (int i = 0; i < I; ++i) {
      dwi[i] += __shfl_xor(dwi[i], 1, 64);
      dwi[i] += __shfl_xor(dwi[i], 2, 64);
      if (part == 0) red[t][i] = dwi[i];
    }
    __syncthreads();
    if (threadIdx.x < I) {
      float s = 0.f;
      for (int k = 0; k < 64; ++k) s += red[k][threadIdx.x];
      A.dWi_part[((size_t)b * gridDim.y + blockIdx.y) * I + threadIdx.x] = s;
    }
  }
  // ---- pass 2, sender q: d aw, d w (sums over the receivers; wgt and d wgt recomputed)
  {
    float ws[D], dws[D], daw = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { ws[d] = A.cs * A.w[((size_t)b * Z + qc) * D + d]; dws[d] = 0.f; }
    const float awq = A.aw[(size_t)b * Z + qc];
    for (int r = part; r < Z; r += 4) {
      float wgt = awq, dwgt = 0.f;
#pragma unroll
      for (int i = 0; i < I; ++i) wgt += A.inv[(((size_t)b * Z + r) * Z + qc) * I + i] * wi[i];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float gd = A.g[((size_t)b * Z + r) * D + d] * invZ;
        dwgt += gd * (A.cr * A.u[((size_t)b * Z + r) * D + d] + ws[d]);
        dws[d] += wgt * gd;
      }
      daw += dwgt;
    }
    daw += __shfl_xor(daw, 1, 64);
    daw += __shfl_xor(daw, 2, 64);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dws[d] += __shfl_xor(dws[d], 1, 64);
      dws[d] += __shfl_xor(dws[d], 2, 64);
    }
    if (ok && part == 0) {
      A.daw[(size_t)b * Z + q] = daw;
#pragma unroll
      for (int d = 0; d < D; ++d) A.dw[((size_t)b * Z + q) * D + d] = A.cs * dws[d];
    }
  }
}

static int ode_vec_check(int B, int Z, int I, int D) {
  if (B <= 0 || Z <= 0 || B > 65535) return ENF_EINVAL;
  if (I < 1 || I > 6 || D < 2 || D > 3) return ENF_EUNSUPPORTED;
  return ENF_OK;
}
#define ODE_VEC_DISPATCH(KERN)                                                                         \
  switch (I * 10 + D) {                                                                                \
    case 12: hipLaunchKernelGGL((KERN<1, 2>), grid, dim3(256), 0, st, A); break;                       \
    case 13: hipLaunchKernelGGL((KERN<1, 3>), grid, dim3(256), 0, st, A); break;                       \
    case 22: hipLaunchKernelGGL((KERN<2, 2>), grid, dim3(256), 0, st, A); break;                       \
    case 23: hipLaunchKernelGGL((KERN<2, 3>), grid, dim3(256), 0, st, A); break;                       \
    case 32: hipLaunchKernelGGL((KERN<3, 2>), grid, dim3(256), 0, st, A); break;                       \
    case 33: hipLaunchKernelGGL((KERN<3, 3>), grid, dim3(256), 0, st, A); break;                       \
    case 42: hipLaunchKernelGGL((KERN<4, 2>), grid, dim3(256), 0, st, A); break;                       \
    case 43: hipLaunchKernelGGL((KERN<4, 3>), grid, dim3(256), 0, st, A); break;                       \
    case 52: hipLaunchKernelGGL((KERN<5, 2>), grid, dim3(256), 0, st, A); break;                       \
    case 53: hipLaunchKernelGGL((KERN<5, 3>), grid, dim3(256), 0, st, A); break;                       \
    case 62: hipLaunchKernelGGL((KERN<6, 2>), grid, dim3(256), 0, st, A); break;                       \
    default: hipLaunchKernelGGL((KERN<6, 3>), grid, dim3(256), 0, st, A); break;                       \
  }

extern "C" int enf_ode_vec_readout_forward(int B, int Z, int I, int D, const float* inv, const float* aw, const float* u,
                                           const float* w, float cr, float cs, const float* Wi, float* out, void* stream) {
  int rc = ode_vec_check(B, Z, I, D);
  if (rc) return rc;
  if (!inv || !aw || !u || !w || !Wi || !out) return ENF_EINVAL;
  OdeVecArgs A{inv, aw, u, w, Wi, nullptr, out, nullptr, nullptr, nullptr, nullptr, nullptr, B, Z, I, D, cr, cs};
  const dim3 grid(B, (Z + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  ODE_VEC_DISPATCH(enf_ode_vec_fwd_kernel)
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

// dWi_part: (B * ceil(Z / 64), I) partial sums, to be added up by the caller (fixed order)
extern "C" int enf_ode_vec_readout_backward(int B, int Z, int I, int D, const float* inv, const float* aw, const float* u,
                                            const float* w, float cr, float cs, const float* Wi, const float* g, float* dinv,
                                            float* daw, float* du, float* dw, float* dWi_part, void* stream) {
  int rc = ode_vec_check(B, Z, I, D);
  if (rc) return rc;
  if (!inv || !aw || !u || !w || !Wi || !g || !dinv || !daw || !du || !dw || !dWi_part) return ENF_EINVAL;
  OdeVecArgs A{inv, aw, u, w, Wi, g, nullptr, dinv, daw, du, dw, dWi_part, B, Z, I, D, cr, cs};
  const dim3 grid(B, (Z + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  ODE_VEC_DISPATCH(enf_ode_vec_bwd_kernel)
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}
