// enf_prologue.hip -- K1 and its backward on the matrix pipe (fp32 `v_mfma_f32_16x16x4_f32`), 16 latents per workgroup.
//
// The per-latent part of the decoder (NEF:220, NEF:56, ECA:93-94 and the logit fold of enf_pack.hip):
//   s = a Ws + bs -> LayerNorm -> an;  k = an Wk + bk;  v0 = an Wv + bv;  u_h = MU_h k_h;  c_h = cvec_h . k_h
// is a chain of small dense products over the B Z latent rows.  The first version (enf_pack.hip: 4 latents per workgroup, one
// output feature per thread, weights streamed from L2 per workgroup) was bound by the latency of that stream -- every workgroup read
// all 392 KB of prologue weights for 4 latents: 100 MB of L2 traffic per call, 20 us forward / 35 us backward for 0.2 GFLOP, three
// times per inner step.  Here a workgroup takes SIXTEEN latents -- the column count of an MFMA tile -- and its eight waves split the
// output features of every layer: each weight element is read once per 16 latents as an A-operand dword (coalesced: the output index
// on the lanes), activations sit in LDS as [feature][latent] (the B operand is one conflict-free ds_read_b32), accumulator tiles go
// back to LDS as the next layer's input.  Same arithmetic (fp32 throughout), same buffers (`an`, `kv`, latent table, `pg`).
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>
#include "enf_layout.h"
#include "enf_launch.h"

namespace {
constexpr int LT = 16;            // latents per workgroup
constexpr int PW = 8;             // waves per workgroup
typedef float pf4 __attribute__((ext_vector_type(4)));

// acc (16 out features i0.. x 16 latents) += sum_k A[k][i0 + i] * sB[k][latent]:  A row-major with `lda` floats between k rows
// (the OUTPUT index contiguous: coalesced), sB = LDS [K][LT].
__device__ __forceinline__ void mm_tile(pf4& acc, const float* __restrict__ A, int lda, int i0, int K, const float* sB, int lane) {
  const int i = lane & 15, kk = lane >> 4;
  const float* ap = A + (size_t)kk * lda + i0 + i;
  const float* bp = sB + kk * LT + i;
  int k0 = 0;
  for (; k0 + 32 <= K; k0 += 32) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = ap[(size_t)(k0 + 4 * u) * lda]; b[u] = bp[(k0 + 4 * u) * LT]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k0 < K; k0 += 4) {
    const bool ok = k0 + kk < K;
    const float a = ok ? ap[(size_t)k0 * lda] : 0.f, b = ok ? bp[k0 * LT] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
}
// as mm_tile with the REDUCTION index contiguous in memory: A[i][k] = M[(i0 + i) * ldm + k] (rows of M are output features; small matrices only)
__device__ __forceinline__ void mm_tile_rows(pf4& acc, const float* __restrict__ M, int ldm, int i0, int imax, int K, const float* sB, int lane) {
  const int i = lane & 15, kk = lane >> 4;
  const bool iok = i0 + i < imax;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const bool ok = iok && k0 + kk < K;
    const float a = ok ? M[(size_t)(i0 + i) * ldm + k0 + kk] : 0.f, b = k0 + kk < K ? sB[(k0 + kk) * LT + i] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
}
// NT tiles at once: acc[t] += sum_{k < K} A_t[k][.] * B_t[k][latent], K a multiple of 32.  `ap[t]` / `bp[t]` already point at this lane's
// element of row k = lane >> 4 (A_t: + out index, `lda` floats per k row; B_t: LDS [k][LT], + latent).  8 k-steps x NT tiles of operands are in
// flight per batch and consecutive MFMAs go to different accumulators -- a lone tile's chain of dependent fp32 MFMAs (40 cycles each)
// with 16 loads in flight was what bound the first form of these kernels.
template <int NT>
__device__ __forceinline__ void mm_tiles(pf4 (&acc)[NT], const float* const (&ap)[NT], const float* const (&bp)[NT], int lda, int K) {
  // Left to itself hipcc sinks every load to just in front of its MFMA -- one load in flight, `s_waitcnt vmcnt(0)` per MFMA -- and runs
  // each tile's eight MFMAs back to back: the fences keep a batch's 16 NT loads together and the accumulators interleaved.
  // (A second register set that requests batch k + 1 before batch k is multiplied measured 10 % slower: round 3)
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[NT][8], b[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[t][u] = ap[t][(size_t)(k0 + 4 * u) * lda]; b[t][u] = bp[t][(k0 + 4 * u) * LT]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u], b[t][u], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
__device__ __forceinline__ float half_sum(float v) {          // sum over the 32 lanes of a half-wave
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct ProArgs {
  const float* p; const float* a; const float* sigma;
  const char* blob; EnfLayout L;
  float* lt; float* an; float* kv;
  int BZ, H, D, C, dp, inv, Dt;
};

__global__ __launch_bounds__(64 * PW) void enf_prologue_mfma_kernel(ProArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = A.D, H = A.H, HD = H * D, C = A.C;
  float* s_a = sm;                          // [C][LT]
  float* s_s = s_a + C * LT;                // [D][LT]   stem output
  float* s_an = s_s + D * LT;               // [D][LT]   LayerNorm output (affine)
  float* s_k = s_an + D * LT;               // [2HD][LT] k | v0
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * LT;
  auto W = [&](size_t off) { return reinterpret_cast<const float*>(A.blob + off); };
  const int stride = enf_lt_stride(H, D);
  for (int t = tid; t < LT * C; t += 64 * PW) {
    const int zz = t / C, c = t % C, r = row0 + zz;
    s_a[c * LT + zz] = r < A.BZ ? A.a[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  const int r_mine = row0 + j;              // this lane's latent in the accumulator layout
  const bool rok = r_mine < A.BZ;
  // ---- stem: s = a Ws + bs (NEF:220)
  for (int t = wave; t < D / 16; t += PW) {
    pf4 acc = *reinterpret_cast<const pf4*>(W(A.L.stem_b) + 16 * t + 4 * q);
    mm_tile(acc, W(A.L.stem_w), D, 16 * t, C, s_a, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) s_s[(16 * t + 4 * q + e) * LT + j] = acc[e];
  }
  __syncthreads();
  // ---- LayerNorm (NEF:56): biased variance, eps 1e-6, statistics over the Dt real features; one half-wave per latent
  {
    const int zz = 2 * wave + (lane >> 5), l32 = lane & 31, r = row0 + zz;
    float v = 0.f;
    for (int d = l32; d < A.Dt; d += 32) v += s_s[d * LT + zz];
    const float mu = half_sum(v) / A.Dt;
    float qv = 0.f;
    for (int d = l32; d < A.Dt; d += 32) { const float t = s_s[d * LT + zz] - mu; qv += t * t; }
    const float rstd = rsqrtf(half_sum(qv) / A.Dt + 1e-6f);
    for (int d = l32; d < D; d += 32) {
      const float sv = s_s[d * LT + zz], xn = (sv - mu) * rstd;
      if (r < A.BZ) {
        A.an[(size_t)r * (2 * D + 2) + d] = sv;
        A.an[(size_t)r * (2 * D + 2) + D + d] = xn;
      }
      s_an[d * LT + zz] = xn * W(A.L.lna_g)[d] + W(A.L.lna_b)[d];
    }
    if (l32 == 0 && r < A.BZ) { A.an[(size_t)r * (2 * D + 2) + 2 * D] = mu; A.an[(size_t)r * (2 * D + 2) + 2 * D + 1] = rstd; }
  }
  __syncthreads();
  // ---- k = an Wk + bk, v0 = an Wv + bv (ECA:93-94): 2 HD output features, a wave's tiles in one pass (shared B operand)
  {
    const int T = 2 * HD / 16;                       // 8, 16 or 32 tiles: 1, 2 or 4 per wave
    auto kv_tiles = [&](auto nt_c) {
      constexpr int NT = decltype(nt_c)::value;
      pf4 acc[NT];
      const float* ap[NT];
      const float* bp[NT];
      int f0[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f0[t] = 16 * (wave + PW * t);
        const bool isv = f0[t] >= HD;
        const int c0 = isv ? f0[t] - HD : f0[t];
        acc[t] = *reinterpret_cast<const pf4*>(W(isv ? A.L.bv : A.L.bk) + c0 + 4 * q);
        ap[t] = W(isv ? A.L.wv : A.L.wk) + (size_t)q * HD + c0 + j;
        bp[t] = s_an + q * LT + j;
      }
      mm_tiles<NT>(acc, ap, bp, HD, D);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool isv = f0[t] >= HD;
#pragma unroll
        for (int e = 0; e < 4; ++e) s_k[(f0[t] + 4 * q + e) * LT + j] = acc[t][e];
        if (rok) {
          *reinterpret_cast<pf4*>(A.kv + (size_t)r_mine * 2 * HD + f0[t] + 4 * q) = acc[t];
          if (isv) *reinterpret_cast<pf4*>(A.lt + (size_t)r_mine * stride + enf_lt_off_v0(H, D) + f0[t] - HD + 4 * q) = acc[t];
        }
      }
    };
    if (T == 32) kv_tiles(std::integral_constant<int, 4>{});
    else if (T == 16) kv_tiles(std::integral_constant<int, 2>{});
    else kv_tiles(std::integral_constant<int, 1>{});
  }
  __syncthreads();
  // ---- u_h = MU_h k_h (mut[h][d][i]: the output index i contiguous)
  {
    const int T = HD / 16;                           // 4, 8 or 16 tiles
    auto u_tiles = [&](auto nt_c) {
      constexpr int NT = decltype(nt_c)::value;
      pf4 acc[NT];
      const float* ap[NT];
      const float* bp[NT];
      int f0[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f0[t] = 16 * (wave + PW * t);
        const int h = f0[t] / D, i0 = f0[t] % D;
        acc[t] = pf4{0.f, 0.f, 0.f, 0.f};
        ap[t] = W(A.L.mut) + (size_t)h * D * D + (size_t)q * D + i0 + j;
        bp[t] = s_k + (h * D + q) * LT + j;
      }
      mm_tiles<NT>(acc, ap, bp, D, D);
      if (rok) {
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<pf4*>(A.lt + (size_t)r_mine * stride + enf_lt_off_u(H, D) + f0[t] + 4 * q) = acc[t];
      }
    };
    if (T == 16) u_tiles(std::integral_constant<int, 2>{});
    else if (T == 8 || wave < T) u_tiles(std::integral_constant<int, 1>{});
  }
  // ---- c_h = cvec_h . k_h: one half-wave per latent
  {
    const int zz = 2 * wave + (lane >> 5), l32 = lane & 31, r = row0 + zz;
    for (int h = 0; h < H; ++h) {
      float sacc = 0.f;
      for (int dd = l32; dd < D; dd += 32) sacc = fmaf(W(A.L.cvec)[h * D + dd], s_k[(h * D + dd) * LT + zz], sacc);
      sacc = half_sum(sacc);
      if (l32 == 0 && r < A.BZ) A.lt[(size_t)r * stride + enf_lt_off_c(H, D) + h] = sacc;
    }
  }
  // ---- pose embed (NEF:214-217) + window coefficient (+ ball / ball_lat: rotation matrix and RFF phases)
  if (tid < LT) {
    const int r = row0 + tid;
    if (r < A.BZ) {
      const float* pp = A.p + (size_t)r * A.dp;
      float qv[4] = {0.f, 0.f, 0.f, 0.f};
      const float sg = A.sigma ? A.sigma[r] : 1.f;
      const bool sphere = A.inv == ENF_INV_LATITUDE_PERIODIC || A.inv == ENF_INV_POLAR_PERIODIC || enf_inv_has_phase(A.inv);
      if (A.inv == ENF_INV_PONITA || A.inv == ENF_INV_PONITA_FULL) { qv[0] = pp[0]; qv[1] = pp[1]; qv[2] = cosf(pp[2]); qv[3] = sinf(pp[2]); }
      else if (sphere) { qv[0] = pp[0]; qv[1] = pp[1]; qv[2] = sinf(pp[1]); qv[3] = cosf(pp[1]); }
      else { for (int i = 0; i < A.dp && i < 3; ++i) qv[i] = pp[i]; }
      const float wc = sphere ? 1.f / (2.f * sg * sg) : 1.f / (sg * sg);
      float* o = A.lt + (size_t)r * stride;
      for (int i = 0; i < 4; ++i) o[enf_lt_off_pose(H, D) + i] = qv[i];
      o[enf_lt_off_wcoef(H, D)] = wc;
      if (enf_inv_has_phase(A.inv)) {
        float lat[2] = {0.f, 0.f};
        if (A.inv == ENF_INV_BALL) {            // R(alpha, beta, gamma), ball.py:76-84; latent-only invariant r_p
          const float ca = cosf(pp[0]), sa = sinf(pp[0]), cb = cosf(pp[1]), sb = sinf(pp[1]), cg = cosf(pp[2]), sg2 = sinf(pp[2]);
          float* R = o + enf_lt_off_ext(H, D);
          R[0] = ca * cb; R[1] = ca * sb * sg2 - sa * cg; R[2] = ca * sb * cg + sa * sg2;
          R[3] = sa * cb; R[4] = sa * sb * sg2 + ca * cg; R[5] = sa * sb * cg - ca * sg2;
          R[6] = -sb;     R[7] = cb * sg2;                R[8] = cb * cg;
          lat[0] = pp[3];
        } else { lat[0] = pp[1]; lat[1] = pp[3]; }      // ball_lat: th_p, r_p
        const float* cq = W(A.L.cphq), *cv = W(A.L.cphv);
        for (int jj = 0; jj < D / 2; ++jj) {            // phase in revolutions (the kernels' sin/cos take 2 pi t)
          o[enf_lt_off_phq(H, D) + jj] = lat[0] * cq[jj] + lat[1] * cq[D / 2 + jj];
          o[enf_lt_off_phv(H, D) + jj] = lat[0] * cv[jj] + lat[1] * cv[D / 2 + jj];
        }
      }
    }
  }
}

struct ProBwdArgs {
  const float* p; const float* sigma; const char* blob; EnfLayout L;
  const float* an; const float* kv; const float* dlt;
  float* dp; float* da; float* dsigma; float* pg;
  int BZ, H, D, C, dp_dim, inv, Dt;
};

// d lt rows (d u | d v0 | d c | d pose | d wcoef) -> d p, d a, d sigma (and, with pg, the operand rows of the weight gradients:
// [d k (HD) | d an (D) | d s (D) | an (D) | d an * xn (D) | d c_h k_h (HD)], enf_train.hip)
__global__ __launch_bounds__(64 * PW) void enf_prologue_bwd_mfma_kernel(ProBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = A.D, H = A.H, HD = H * D, C = A.C, PGW = 2 * HD + 4 * D;
  float* s_du = sm;                    // [HD][LT]   d u
  float* s_dk = s_du + HD * LT;        // [2HD][LT]  d k | d v0
  float* s_dan = s_dk + 2 * HD * LT;   // [D][LT]    d an, then d xn, then d s
  float* s_dc = s_dan + D * LT;        // [H][LT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4, row0 = blockIdx.x * LT;
  const int stride = enf_lt_stride(H, D);
  auto W = [&](size_t off) { return reinterpret_cast<const float*>(A.blob + off); };
  for (int t = tid; t < LT * HD; t += 64 * PW) {
    const int zz = t / HD, f = t % HD, r = row0 + zz;
    const float* g = A.dlt + (size_t)(r < A.BZ ? r : 0) * stride;
    s_du[f * LT + zz] = r < A.BZ ? g[enf_lt_off_u(H, D) + f] : 0.f;
    s_dk[(HD + f) * LT + zz] = r < A.BZ ? g[enf_lt_off_v0(H, D) + f] : 0.f;
  }
  if (tid < LT * H) {
    const int zz = tid / H, h = tid % H, r = row0 + zz;
    s_dc[h * LT + zz] = r < A.BZ ? A.dlt[(size_t)r * stride + enf_lt_off_c(H, D) + h] : 0.f;
  }
  __syncthreads();
  const int r_mine = row0 + j;
  const bool rok = r_mine < A.BZ;
  // ---- d k_h[d] = sum_i MU_h[i][d] d u_h[i] + cvec_h[d] d c_h   (mu[h][i][d]: the output index d contiguous)
  {
    const int T = HD / 16;
    auto dk_tiles = [&](auto nt_c) {
      constexpr int NT = decltype(nt_c)::value;
      pf4 acc[NT];
      const float* ap[NT];
      const float* bp[NT];
      int f0[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f0[t] = 16 * (wave + PW * t);
        const int h = f0[t] / D, d0 = f0[t] % D;
        acc[t] = pf4{0.f, 0.f, 0.f, 0.f};
        ap[t] = W(A.L.mu) + (size_t)h * D * D + (size_t)q * D + d0 + j;
        bp[t] = s_du + (h * D + q) * LT + j;
      }
      mm_tiles<NT>(acc, ap, bp, D, D);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int h = f0[t] / D;
        const float dc = s_dc[h * LT + j];
        const pf4 cv = *reinterpret_cast<const pf4*>(W(A.L.cvec) + f0[t] + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[t][e] = fmaf(cv[e], dc, acc[t][e]); s_dk[(f0[t] + 4 * q + e) * LT + j] = acc[t][e]; }
        if (A.pg && rok) {
          float* pr = A.pg + (size_t)r_mine * PGW;
          const pf4 kq = *reinterpret_cast<const pf4*>(A.kv + (size_t)r_mine * 2 * HD + f0[t] + 4 * q);
          *reinterpret_cast<pf4*>(pr + f0[t] + 4 * q) = acc[t];
          *reinterpret_cast<pf4*>(pr + HD + 4 * D + f0[t] + 4 * q) = pf4{dc * kq[0], dc * kq[1], dc * kq[2], dc * kq[3]};
        }
      }
    };
    if (T == 16) dk_tiles(std::integral_constant<int, 2>{});
    else if (T == 8 || wave < T) dk_tiles(std::integral_constant<int, 1>{});
  }
  __syncthreads();
  // ---- d an[d] = sum_f Wk[d][f] d k[f] + Wv[d][f] d v0[f]   (wkT / wvT: [f][d]); d xn = d an * g.  One out-tile per wave, its K = 2 HD
  // split over four accumulators (two halves of each matrix)
  if (wave < D / 16) {
    const int t = wave, hk = HD / 2;
    pf4 acc[4];
    const float* ap[4];
    const float* bp[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      acc[v] = pf4{0.f, 0.f, 0.f, 0.f};
      ap[v] = W(v < 2 ? A.L.wkt : A.L.wvt) + (size_t)((v & 1) * hk + q) * D + 16 * t + j;
      bp[v] = s_dk + ((v < 2 ? 0 : HD) + (v & 1) * hk + q) * LT + j;
    }
    mm_tiles<4>(acc, ap, bp, D, hk);
    pf4 sum;
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    const pf4 g = *reinterpret_cast<const pf4*>(W(A.L.lna_g) + 16 * t + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) s_dan[(16 * t + 4 * q + e) * LT + j] = sum[e] * g[e];
    if (A.pg && rok) {
      float* pr = A.pg + (size_t)r_mine * PGW + HD;
      const pf4 xn = *reinterpret_cast<const pf4*>(A.an + (size_t)r_mine * (2 * D + 2) + D + 16 * t + 4 * q);
      const pf4 b = *reinterpret_cast<const pf4*>(W(A.L.lna_b) + 16 * t + 4 * q);
      *reinterpret_cast<pf4*>(pr + 16 * t + 4 * q) = sum;
      *reinterpret_cast<pf4*>(pr + 2 * D + 16 * t + 4 * q) = pf4{xn[0] * g[0] + b[0], xn[1] * g[1] + b[1], xn[2] * g[2] + b[2], xn[3] * g[3] + b[3]};
      *reinterpret_cast<pf4*>(pr + 3 * D + 16 * t + 4 * q) = pf4{sum[0] * xn[0], sum[1] * xn[1], sum[2] * xn[2], sum[3] * xn[3]};
    }
  }
  __syncthreads();
  // ---- LayerNorm backward: d s = rstd (d xn - mean(d xn) - xn mean(d xn xn)); one half-wave per latent
  {
    const int zz = 2 * wave + (lane >> 5), l32 = lane & 31, r = row0 + zz;
    const int rr = r < A.BZ ? r : A.BZ - 1;
    const float* anr = A.an + (size_t)rr * (2 * D + 2);
    float v1 = 0.f, v2 = 0.f;
    for (int d = l32; d < D; d += 32) { const float g = s_dan[d * LT + zz]; v1 += g; v2 += g * anr[D + d]; }
    const float m1 = half_sum(v1) / A.Dt, m2 = half_sum(v2) / A.Dt;
    const float rstd = anr[2 * D + 1];
    for (int d = l32; d < D; d += 32) {
      const float ds = rstd * (s_dan[d * LT + zz] - m1 - anr[D + d] * m2);
      s_dan[d * LT + zz] = ds;
      if (A.pg && r < A.BZ) A.pg[(size_t)r * PGW + HD + D + d] = ds;
    }
  }
  __syncthreads();
  // ---- d a[c] = sum_d Ws[c][d] d s[d]
  for (int t = wave; t < (C + 15) / 16; t += PW) {
    pf4 acc = {0.f, 0.f, 0.f, 0.f};
    mm_tile_rows(acc, W(A.L.stem_w), D, 16 * t, C, D, s_dan, lane);
    if (rok) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (16 * t + 4 * q + e < C) A.da[(size_t)r_mine * C + 16 * t + 4 * q + e] = acc[e];
    }
  }
  if (tid < LT) {
    const int r = row0 + tid;
    if (r < A.BZ) {
      const float* g = A.dlt + (size_t)r * stride + enf_lt_off_pose(H, D);
      const float* pp = A.p + (size_t)r * A.dp_dim;
      float* o = A.dp + (size_t)r * A.dp_dim;
      const bool sph = A.inv == ENF_INV_LATITUDE_PERIODIC || A.inv == ENF_INV_POLAR_PERIODIC || enf_inv_has_phase(A.inv);
      if (A.inv == ENF_INV_PONITA || A.inv == ENF_INV_PONITA_FULL) { o[0] = g[0]; o[1] = g[1]; o[2] = -sinf(pp[2]) * g[2] + cosf(pp[2]) * g[3]; }
      else if (sph) { o[0] = g[0]; o[1] = g[1] + cosf(pp[1]) * g[2] - sinf(pp[1]) * g[3]; }
      else { for (int i = 0; i < A.dp_dim && i < 3; ++i) o[i] = g[i]; }
      if (enf_inv_has_phase(A.inv)) {
        const float* e = A.dlt + (size_t)r * stride + enf_lt_off_ext(H, D);     // d R (9) | d(latent-only invariants) (2)
        if (A.inv == ENF_INV_BALL) {
          const float ca = cosf(pp[0]), sa = sinf(pp[0]), cb = cosf(pp[1]), sb = sinf(pp[1]), cg = cosf(pp[2]), sg2 = sinf(pp[2]);
          o[0] += e[0] * (-sa * cb) + e[1] * (-sa * sb * sg2 - ca * cg) + e[2] * (-sa * sb * cg + ca * sg2) +
                  e[3] * (ca * cb) + e[4] * (ca * sb * sg2 - sa * cg) + e[5] * (ca * sb * cg + sa * sg2);
          o[1] += e[0] * (-ca * sb) + e[1] * (ca * cb * sg2) + e[2] * (ca * cb * cg) + e[3] * (-sa * sb) + e[4] * (sa * cb * sg2) +
                  e[5] * (sa * cb * cg) + e[6] * (-cb) + e[7] * (-sb * sg2) + e[8] * (-sb * cg);
          o[2] = e[1] * (ca * sb * cg + sa * sg2) + e[2] * (-ca * sb * sg2 + sa * cg) + e[4] * (sa * sb * cg - ca * sg2) +
                 e[5] * (-sa * sb * sg2 - ca * cg) + e[7] * (cb * cg) + e[8] * (-cb * sg2);
          o[3] = e[9];
        } else { o[1] += e[9]; o[2] = 0.f; o[3] = e[10]; }
      }
      const float sg = A.sigma ? A.sigma[r] : 1.f;
      const float dwc = A.dlt[(size_t)r * stride + enf_lt_off_wcoef(H, D)];
      A.dsigma[r] = (sph ? -1.f : -2.f) / (sg * sg * sg) * dwc;
    }
  }
}
}  // namespace

extern "C" int enf_launch_prologue_mfma(const EnfDims& m, const EnfLayout& L, const char* blob, const float* p, const float* a,
                                        const float* sigma, float* lt, float* an, float* kv, hipStream_t st) {
  ProArgs A;
  A.p = p; A.a = a; A.sigma = sigma; A.blob = blob; A.L = L; A.lt = lt; A.an = an; A.kv = kv;
  A.BZ = m.B * m.Z; A.H = m.H; A.D = m.D; A.C = m.C; A.dp = m.dp; A.inv = m.inv; A.Dt = m.Dt;
  const size_t smem = sizeof(float) * LT * (m.C + 2 * m.D + 2 * m.HD);
  static EnfAttrBits attr{0};
  if (!enf_lds_attr(reinterpret_cast<const void*>(enf_prologue_mfma_kernel), 160 * 1024, attr)) return ENF_ELAUNCH;
  hipLaunchKernelGGL(enf_prologue_mfma_kernel, dim3((A.BZ + LT - 1) / LT), dim3(64 * PW), smem, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

extern "C" int enf_launch_prologue_bwd_mfma(const EnfDims& m, const EnfLayout& L, const char* blob, const float* p, const float* sigma,
                                            const float* an, const float* kv, const float* dlt, float* dp, float* da, float* dsigma,
                                            float* pg, hipStream_t st) {
  ProBwdArgs A;
  A.p = p; A.sigma = sigma; A.blob = blob; A.L = L; A.an = an; A.kv = kv; A.dlt = dlt;
  A.dp = dp; A.da = da; A.dsigma = dsigma; A.pg = pg;
  A.BZ = m.B * m.Z; A.H = m.H; A.D = m.D; A.C = m.C; A.dp_dim = m.dp; A.inv = m.inv; A.Dt = m.Dt;
  const size_t smem = sizeof(float) * LT * (3 * m.HD + m.D + m.H);
  static EnfAttrBits attr{0};
  if (!enf_lds_attr(reinterpret_cast<const void*>(enf_prologue_bwd_mfma_kernel), 160 * 1024, attr)) return ENF_ELAUNCH;
  hipLaunchKernelGGL(enf_prologue_bwd_mfma_kernel, dim3((A.BZ + LT - 1) / LT), dim3(64 * PW), smem, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}
