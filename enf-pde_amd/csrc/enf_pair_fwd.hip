// enf_pair_fwd.hip -- K2: fused per-pair chain + cross-attention over the latent set.
//
// One workgroup (4 waves, one per SIMD) owns a tile of 32 queries of one signal; its waves
// split the Z latents (wave w takes z = w, w+4, ..).  For each (32 queries) x (one latent) the
// wave runs, entirely in registers and in the transposed "acc layout" of enf_device.h:
//   invariant (INV/*)                -> inv (I<=4)                              ECA:86
//   t = coeff^T inv (fp32 MFMA), [sin,cos]                                      RFF:86-93
//   h1 = relu(W1q^T e + b)           (query RFFNet layer)                       RFF:63-64
//   logit_h = h1.u_h + c_h + window  (RFF linear_final, inv_emb_to_q and the q.k dot folded
//                                     into the per-latent vector u_h)            RFF:46, ECA:92,134,139
//   g1 = relu(W1v^T e_v + b); f = gelu(AF^T g1 + b); n = LayerNorm(f)           RFF:63-64,46; ECA:17-19
//   [gamma_h; beta_h] = AGB^T n + b ; v_h = v0_h (1+gamma_h) + beta_h           ECA:20,115-121
//   g = gelu(AM^T v_h + b); (mu, rstd) = LN stats of g                          ECA:122 -> ECA:17-19
//   online softmax over z of logit_h; ybar_h += softmax * (g - mu) * rstd       ECA:141-144
// The mixer's LayerNorm affine and Dense_1, attn.out_proj and the block FFN's Dense_0 are
// linear in the softmax-weighted sum and are applied once per query by the tail kernel.
// Weight panels stream L2 -> LDS through a 2-deep ring shared by the 4 waves.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_device.h"
#include "enf_pair_common.h"

#ifndef USE_IDMFMA
#define USE_IDMFMA 0
#endif
struct PairFwdArgs {
  const float* x; long long x_bstride;
  const float* lt; const char* blob; EnfLayout L;
  float* ybar; float* lse;
  int B, N, Z, dx, inv, use_window;
};

template <int D, int H, bool BF16> struct PairSmem {
  static constexpr int RING = 0;                                   // 2 slots
  static constexpr int CONSTS = RING + 2 * STAGE_MAX;              // bq1 bv1 bf bm (D each) bgb (2HD) acq acv
  static constexpr int N_CONST = 4 * D + 2 * H * D + 2 * (D / 64) * 128;
  static constexpr int ZVEC = CONSTS + 4 * N_CONST;                // 4 waves x 2*H*D floats
  static constexpr int XCH = ZVEC + 4 * 4 * 2 * H * D;             // 4 waves x H x 3 x 32
  static constexpr int TOTAL = XCH + 4 * 4 * H * 3 * 32;
  static constexpr int COMBINE_BYTES = H * (D / 32) * 16 * 64 * 4;
  static_assert(COMBINE_BYTES <= 2 * STAGE_MAX, "combine buffer must fit in the ring");
};

template <int D, int H, bool BF16>
__global__ __launch_bounds__(256, 1) void enf_pair_fwd_kernel(PairFwdArgs A) {
  using Cfg = PairCfg<D, BF16>;
  using SM = PairSmem<D, H, BF16>;
  constexpr int KB = Cfg::KB;
  constexpr int ST_DD = Cfg::DD::STAGE, ST_GB = Cfg::GB::STAGE, PANEL_GB = Cfg::GB::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem + SM::RING;
  float* cst = reinterpret_cast<float*>(smem + SM::CONSTS);
  float* c_bq1 = cst, *c_bv1 = cst + D, *c_bf = cst + 2 * D, *c_bm = cst + 3 * D, *c_bgb = cst + 4 * D;
  float* c_acq = c_bgb + 2 * H * D, *c_acv = c_acq + (D / 64) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  float* zv = reinterpret_cast<float*>(smem + SM::ZVEC) + wave * 2 * H * D;
  float* xch = reinterpret_cast<float*>(smem + SM::XCH);
  const int b = blockIdx.y, n0 = blockIdx.x * 32;
  const int n = min(n0 + col, A.N - 1);
  const char* blob = A.blob;
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };

  // ---- constants -> LDS
  for (int i = tid; i < D; i += 256) { c_bq1[i] = G(A.L.bq1)[i]; c_bv1[i] = G(A.L.bv1)[i]; c_bf[i] = G(A.L.bf)[i]; c_bm[i] = G(A.L.bm)[i]; }
  for (int i = tid; i < 2 * H * D; i += 256) c_bgb[i] = G(A.L.bgb)[i];
  for (int i = tid; i < (D / 64) * 128; i += 256) { c_acq[i] = G(A.L.acq)[i]; c_acv[i] = G(A.L.acv)[i]; }

  // ---- this lane's query
  QueryPt q;
  {
    const float* xp = A.x + (size_t)b * A.x_bstride + (size_t)n * A.dx;
    q.x0 = xp[0]; q.x1 = A.dx > 1 ? xp[1] : 0.f; q.x2 = A.dx > 2 ? xp[2] : 0.f;
    q.sx = 0.f; q.cx = 0.f;
    if (A.inv == ENF_INV_LATITUDE_PERIODIC || A.inv == ENF_INV_POLAR_PERIODIC) { q.sx = sinf(q.x1); q.cx = cosf(q.x1); }
  }

  const unsigned pQ1 = (unsigned)A.L.aq1, pV1 = (unsigned)A.L.av1, pF = (unsigned)A.L.af, pGB = (unsigned)A.L.agb, pM = (unsigned)A.L.am;

  Pipe P;
  P.cur = 0;
  P.rs = make_blob_rsrc(blob, (unsigned)A.L.total);
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  stage_issue<ST_DD>(P.rs, pQ1, ring, P.wave, lane);
  stage_wait();
  __syncthreads();

  float sm_m[H], sm_l[H], sm_c[H];
  f32x16 Y[H][KB];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    sm_m[h] = -INFINITY; sm_l[h] = 0.f; sm_c[h] = 0.f;
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[h][k][r] = 0.f;
  }

  // identity A operand matching make_frags' k order: A[i][k] = 1 iff B row k is feature row i
  bf16x8 idf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) idf[s][j] = (__bf16)((col == 16 * s + 8 * (j >> 2) + 4 * half + (j & 3)) ? 1.0f : 0.0f);

  const int ltstride = enf_lt_stride(H, D);
  const int iters = (A.Z + 3) / 4;
  for (int it = 0; it < iters; ++it) {
    const int z = it * 4 + wave;
    const bool active = z < A.Z;
    const float* ltrow = A.lt + ((size_t)b * A.Z + (active ? z : A.Z - 1)) * ltstride;
    // per-latent vectors u | v0 -> wave-private LDS
#pragma unroll
    for (int i = lane * 4; i < 2 * H * D; i += 256)
      *reinterpret_cast<f32x4*>(zv + i) = *reinterpret_cast<const f32x4*>(ltrow + i);
    const f32x4 pz = *reinterpret_cast<const f32x4*>(ltrow + enf_lt_off_pose(H, D));
    const float wcoef = ltrow[enf_lt_off_wcoef(H, D)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float inv[4], win;
    pair_invariant<BF16>(A.inv, A.dx, q, pz, wcoef, A.use_window, inv, win);

    float logit[H];
    Frags<BF16, KB> F;
    {  // ---------------- query branch
      f32x16 E[KB];
      rff_embed<D, BF16>(E, inv, c_acq, lane, half);
      make_frags<BF16, KB>(F, E);
      f32x16 acc[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) load_rowvec(acc[k], c_bq1, k, half);
      panel_gemm<KB, KB, BF16, ST_DD>(acc, F, P, ring, pQ1, pV1, active, tid, lane);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          f32x16 u;
          load_rowvec(u, zv + h * D, k, half);
#pragma unroll
          for (int r = 0; r < 16; ++r) s = fmaf(fmaxf(acc[k][r], 0.f), u[r], s);
        }
        logit[h] = xhalf_sum(s) + ltrow[enf_lt_off_c(H, D) + h] + win;
      }
    }
    {  // ---------------- value branch: RFFNet layer, folded (linear_final . Dense_0), gelu, LN
      f32x16 E[KB];
      rff_embed<D, BF16>(E, inv, c_acv, lane, half);
      make_frags<BF16, KB>(F, E);
      f32x16 acc[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) load_rowvec(acc[k], c_bv1, k, half);
      panel_gemm<KB, KB, BF16, ST_DD>(acc, F, P, ring, pV1, pF, active, tid, lane);
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = fmaxf(acc[k][r], 0.f);
      make_frags<BF16, KB>(F, acc);
#pragma unroll
      for (int k = 0; k < KB; ++k) load_rowvec(acc[k], c_bf, k, half);
      panel_gemm<KB, KB, BF16, ST_GB>(acc, F, P, ring, pF, pGB, active, tid, lane);
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = gelu_f(acc[k][r]);
      float mu, rstd;
      ln_stats<KB>(acc, mu, rstd);
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = (acc[k][r] - mu) * rstd;
      make_frags<BF16, KB>(F, acc);   // F = normalised f, shared by all heads' gamma/beta panels
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      f32x16 v[KB];
      gb_panel<D, BF16, ST_DD>(v, F, P, ring, pGB + h * PANEL_GB, pM, active, c_bgb + 2 * h * D, zv + H * D + h * D,
                               tid, lane, half);
      Frags<BF16, KB> FV;
      make_frags<BF16, KB>(FV, v);
#pragma unroll
      for (int k = 0; k < KB; ++k) load_rowvec(v[k], c_bm, k, half);
      if (h + 1 < H) panel_gemm<KB, KB, BF16, ST_GB>(v, FV, P, ring, pM, pGB + (h + 1) * PANEL_GB, active, tid, lane);
      else panel_gemm<KB, KB, BF16, ST_DD>(v, FV, P, ring, pM, it + 1 < iters ? pQ1 : NO_STAGE, active, tid, lane);
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[k][r] = gelu_f(v[k][r]);
      float mu, rstd;
      ln_stats<KB>(v, mu, rstd);
      if (active) {
        // softmax over this wave's latents (ECA:141-144) against a per-column reference logit
        // sm_m (the first logit seen; fp32 accumulators need no running max).  The rare
        // "logit far above the reference" case rescales the accumulators (wave-uniform branch).
        if (it == 0) sm_m[h] = logit[h];
        const bool far = logit[h] - sm_m[h] > 40.0f;
        if (__any(far)) {
          const float alpha = far ? __expf(sm_m[h] - logit[h]) : 1.0f;
          sm_l[h] *= alpha; sm_c[h] *= alpha;
          if (far) sm_m[h] = logit[h];
#pragma unroll
          for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[h][k][r] *= alpha;
        }
        const float pe = __expf(logit[h] - sm_m[h]);
        const float w = pe * rstd;
        sm_l[h] += pe;
        sm_c[h] = fmaf(w, mu, sm_c[h]);
        if constexpr (BF16 && USE_IDMFMA) {
          // Y += w*g through the matrix pipe (identity A operand): the accumulators stay in the
          // MFMA register file instead of round-tripping through the VALU
#pragma unroll
          for (int k = 0; k < KB; ++k) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              bf16x8 pf;
#pragma unroll
              for (int j = 0; j < 8; ++j) pf[j] = (__bf16)(w * v[k][8 * s + j]);
              Y[h][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf[s], pf, Y[h][k], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[h][k][r] = fmaf(w, v[k][r], Y[h][k][r]);
        }
      }
    }
  }

  // ---- combine the 4 waves' partial softmax states (all staging is finished: ring is free)
  if (half == 0) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      xch[((wave * H + h) * 3 + 0) * 32 + col] = sm_m[h];
      xch[((wave * H + h) * 3 + 1) * 32 + col] = sm_l[h];
      xch[((wave * H + h) * 3 + 2) * 32 + col] = sm_c[h];
    }
  }
  __syncthreads();
  float Ltot[H], Ctot[H], mstar[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float ms = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) ms = fmaxf(ms, xch[((w * H + h) * 3 + 0) * 32 + col]);
    float L = 0.f, C = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float aw = __expf(xch[((w * H + h) * 3 + 0) * 32 + col] - ms);
      L = fmaf(aw, xch[((w * H + h) * 3 + 1) * 32 + col], L);
      C = fmaf(aw, xch[((w * H + h) * 3 + 2) * 32 + col], C);
    }
    mstar[h] = ms; Ltot[h] = L; Ctot[h] = C;
    const float sc = __expf(sm_m[h] - ms) / L;
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[h][k][r] *= sc;
  }
  float* cb = reinterpret_cast<float*>(ring);
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int k = 0; k < KB; ++k)
#pragma unroll
          for (int r = 0; r < 16; ++r) cb[((h * KB + k) * 16 + r) * 64 + lane] = Y[h][k][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int k = 0; k < KB; ++k)
#pragma unroll
          for (int r = 0; r < 16; ++r) Y[h][k][r] += cb[((h * KB + k) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0 && n0 + col < A.N) {
    float* yo = A.ybar + ((size_t)b * A.N + n0 + col) * (H * D);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float cs = Ctot[h] / Ltot[h];
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
          o[0] = Y[h][k][4 * g + 0] - cs; o[1] = Y[h][k][4 * g + 1] - cs;
          o[2] = Y[h][k][4 * g + 2] - cs; o[3] = Y[h][k][4 * g + 3] - cs;
          *reinterpret_cast<f32x4*>(yo + h * D + 32 * k + 8 * g + 4 * half) = o;
        }
      if (half == 0) A.lse[((size_t)b * A.N + n0 + col) * H + h] = mstar[h] + __logf(Ltot[h]);
    }
  }
}

template <int D, int H, bool BF16>
static int launch_pair_fwd(const PairFwdArgs& A, hipStream_t st) {
  using SM = PairSmem<D, H, BF16>;
  auto kern = enf_pair_fwd_kernel<D, H, BF16>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SM::TOTAL) != hipSuccess)
      return ENF_ELAUNCH;
    attr_set = true;
  }
  dim3 grid((A.N + 31) / 32, A.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), SM::TOTAL, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

extern "C" int enf_launch_pair_fwd(const EnfDims& m, const EnfLayout& L, const char* blob, const float* x, long long x_bstride,
                                   const float* lt, float* ybar, float* lse, hipStream_t st) {
  PairFwdArgs A;
  A.x = x; A.x_bstride = x_bstride; A.lt = lt; A.blob = blob; A.L = L; A.ybar = ybar; A.lse = lse;
  A.B = m.B; A.N = m.N; A.Z = m.Z; A.dx = m.dx; A.inv = m.inv; A.use_window = m.use_window;
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) return m.bf16 ? launch_pair_fwd<DD, HH, true>(A, st) : launch_pair_fwd<DD, HH, false>(A, st);
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
