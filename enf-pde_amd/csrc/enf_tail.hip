// enf_tail.hip -- per-query tail of the decoder, forward and backward.
//
// Input: ybar (B,N,H*D) = softmax-weighted sum of the mixer's normalised gelu output (K2).
// Because softmax weights sum to 1, everything linear after ECA:144 is applied once per query:
//   WB  = blockdiag(diag(LN.scale) Dense_1 [mixer, ECA:19-20]) . attn.out_proj (ECA:150) . ffn.Dense_0 (ECA:17)
//   gelu, LayerNorm (ECA:18-19), WF1 = diag(LN.scale) ffn.Dense_1 (ECA:20)            [block FFN, NEF:66]
//   gelu (NEF:230), out_proj: Dense, gelu, Dense, gelu, Dense (NEF:196-202,233)
// One wave = 16 queries, transposed chain as in enf_device.h; 8 waves share the LDS weight ring.
// The backward kernel recomputes this forward (stashing pre-activations in the workspace),
// then runs dX = W dY through the transposed panels and emits d(ybar) and
// delta[n,h] = sum_d d(ybar)[n,h,d] * ybar[n,h,d]  (the softmax-backward row constant).
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_device.h"

struct TailArgs {
  const float* ybar; const char* blob; EnfLayout L;
  float* out;            // forward: (B*N, O)
  const float* dout;     // backward
  float* dybar; float* delta; float* act;   // backward outputs + scratch (B*N x (2HD + 2D + 2))
  int NQ, O;             // NQ = B*N queries
  float inv_hd;          // 1 / (H * true num_hidden)
};

// row-per-query global <-> acc layout (feature f = 16 t + 4 quad + i lives in tile t register i)
template <int NT> DEV void load_rows(f32x4 (&X)[NT], const float* row, int quad) {
#pragma unroll
  for (int t = 0; t < NT; ++t) X[t] = *reinterpret_cast<const f32x4*>(row + 16 * t + 4 * quad);
}
template <int NT> DEV void store_rows(const f32x4 (&X)[NT], float* row, int quad) {
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(row + 16 * t + 4 * quad) = X[t];
}
template <int NT> DEV void zero_tiles(f32x4 (&X)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) X[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int D, int H, bool BF16> struct TailCfg {
  static constexpr int KB = D / 32, KBH = H * D / 32, NT = D / 16, NTH = H * D / 16, HD = H * D;
  static constexpr int ACT = 2 * HD + 2 * D + 2;   // a_B | a_F1 | a_O0 | a_O2 | mu | rstd
  static constexpr int ST_TB = PanelCfg<KBH, NTH, BF16>::STAGE;   // HD x HD
  static constexpr int ST_O0 = PanelCfg<KBH, NT, BF16>::STAGE;    // out D, in HD
  static constexpr int ST_O2 = PanelCfg<KB, NT, BF16>::STAGE;     // D x D
  static constexpr int ST_O4 = PanelCfg<KB, 2, BF16>::STAGE;      // out 32 (O padded), in D
  static constexpr int ST_G4 = PanelCfg<1, NT, BF16>::STAGE;      // gto4: out D, in 32
  static constexpr int ST_G0 = PanelCfg<KB, NTH, BF16>::STAGE;    // gto0: out HD, in D
  static constexpr int SMEM = 2 * STAGE_MAX + 4 * (2 * HD + 2 * D + 32);
};

// Forward chain for this wave's 16 queries.  o4 = the (padded) 32 network outputs.
// SAVE: stash pre-activations + LN stats to `act` (row per query) for the backward chain.
template <int D, int H, bool BF16, bool SAVE, int NEXT_BYTES>
DEV void tail_forward(f32x4 (&o4)[2], const float* yrow, float* act, const EnfLayout& L, const float* cst, Pipe& P,
                      char* ring, unsigned next, int lane, int quad, float inv_hd) {
  using T = TailCfg<D, H, BF16>;
  constexpr int KB = T::KB, KBH = T::KBH, NT = T::NT, NTH = T::NTH, HD = T::HD;
  const float* c_bB = cst, *c_bF1 = cst + HD, *c_bO0 = cst + 2 * HD, *c_bO2 = cst + 2 * HD + D, *c_bO4 = cst + 2 * HD + 2 * D;
  Frags<BF16, KBH> FH;
  f32x4 a[NTH];
  load_rows<NTH>(a, yrow, quad);
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int t = 0; t < NTH; ++t) a[t] = rowvec(c_bB, t, quad);
  panel_gemm<KBH, NTH, BF16, T::ST_TB>(a, FH, P, ring, (unsigned)L.atb, (unsigned)L.atf1, true, lane);
  if (SAVE) store_rows<NTH>(a, act, quad);
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[t][i] = gelu_f(a[t][i]);
  float mu, rstd;
  ln_stats<NTH>(a, mu, rstd, inv_hd);
  if (SAVE && quad == 0) { act[T::ACT - 2] = mu; act[T::ACT - 1] = rstd; }
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[t][i] = (a[t][i] - mu) * rstd;
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int t = 0; t < NTH; ++t) a[t] = rowvec(c_bF1, t, quad);
  panel_gemm<KBH, NTH, BF16, T::ST_O0>(a, FH, P, ring, (unsigned)L.atf1, (unsigned)L.ato0, true, lane);
  if (SAVE) store_rows<NTH>(a, act + HD, quad);
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[t][i] = gelu_f(a[t][i]);       // NEF:230
  make_frags<BF16, KBH>(FH, a);
  f32x4 c[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) c[t] = rowvec(c_bO0, t, quad);
  panel_gemm<KBH, NT, BF16, T::ST_O2>(c, FH, P, ring, (unsigned)L.ato0, (unsigned)L.ato2, true, lane);
  if (SAVE) store_rows<NT>(c, act + 2 * HD, quad);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) c[t][i] = gelu_f(c[t][i]);
  Frags<BF16, KB> FD;
  make_frags<BF16, KB>(FD, c);
#pragma unroll
  for (int t = 0; t < NT; ++t) c[t] = rowvec(c_bO2, t, quad);
  panel_gemm<KB, NT, BF16, T::ST_O4>(c, FD, P, ring, (unsigned)L.ato2, (unsigned)L.ato4, true, lane);
  if (SAVE) store_rows<NT>(c, act + 2 * HD + D, quad);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) c[t][i] = gelu_f(c[t][i]);
  make_frags<BF16, KB>(FD, c);
  o4[0] = rowvec(c_bO4, 0, quad);
  o4[1] = rowvec(c_bO4, 1, quad);
  panel_gemm<KB, 2, BF16, NEXT_BYTES>(o4, FD, P, ring, (unsigned)L.ato4, next, true, lane);
}

template <int D, int H, bool BF16>
DEV void tail_consts(float* cst, const char* blob, const EnfLayout& L, int tid) {
  constexpr int HD = H * D;
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };
  for (int i = tid; i < HD; i += NTHREADS) { cst[i] = G(L.bB)[i]; cst[HD + i] = G(L.bF1)[i]; }
  for (int i = tid; i < D; i += NTHREADS) { cst[2 * HD + i] = G(L.bO0)[i]; cst[2 * HD + D + i] = G(L.bO2)[i]; }
  for (int i = tid; i < 32; i += NTHREADS) cst[2 * HD + 2 * D + i] = G(L.bO4)[i];
}

template <int D, int H, bool BF16>
__global__ __launch_bounds__(NTHREADS, 2) void enf_tail_fwd_kernel(TailArgs A) {
  using T = TailCfg<D, H, BF16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + 2 * STAGE_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, quad = lane >> 4;
  const int q0 = blockIdx.x * (16 * NWAVES) + wave * 16;
  const int qi = min(q0 + col, A.NQ - 1);
  tail_consts<D, H, BF16>(cst, A.blob, A.L, tid);
  Pipe P;
  P.rs = make_blob_rsrc(A.blob, (unsigned)A.L.total);
  P.rs2 = P.rs;
  first_stage<T::ST_TB>(P, ring, (unsigned)A.L.atb, wave, lane);
  f32x4 o4[2];
  tail_forward<D, H, BF16, false, 1024>(o4, A.ybar + (size_t)qi * T::HD, nullptr, A.L, cst, P, ring, NO_STAGE, lane, quad, A.inv_hd);
  if (q0 + col < A.NQ) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = 16 * t + 4 * quad + i;
        if (o < A.O) A.out[(size_t)(q0 + col) * A.O + o] = o4[t][i];
      }
  }
}

template <int D, int H, bool BF16>
__global__ __launch_bounds__(NTHREADS, 2) void enf_tail_bwd_kernel(TailArgs A) {
  using T = TailCfg<D, H, BF16>;
  constexpr int KB = T::KB, KBH = T::KBH, NT = T::NT, NTH = T::NTH, HD = T::HD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + 2 * STAGE_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, quad = lane >> 4;
  const int q0 = blockIdx.x * (16 * NWAVES) + wave * 16;
  const bool qvalid = q0 + col < A.NQ;
  const int qi = min(q0 + col, A.NQ - 1);
  // clamped (duplicate) queries write the same scratch values: benign
  float* act = A.act + (size_t)qi * T::ACT;
  const float* yrow = A.ybar + (size_t)qi * HD;
  tail_consts<D, H, BF16>(cst, A.blob, A.L, tid);
  Pipe P;
  P.rs = make_blob_rsrc(A.blob, (unsigned)A.L.total);
  P.rs2 = P.rs;
  first_stage<T::ST_TB>(P, ring, (unsigned)A.L.atb, wave, lane);
  f32x4 o4[2];
  tail_forward<D, H, BF16, true, T::ST_G4>(o4, yrow, act, A.L, cst, P, ring, (unsigned)A.L.gto4, lane, quad, A.inv_hd);
  // the pre-activations this lane stored are re-read by this lane only (same addresses)

  // ---- backward chain
  f32x4 g0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = 16 * t + 4 * quad + i;
      g0[t][i] = (o < A.O && qvalid) ? A.dout[(size_t)qi * A.O + o] : 0.f;
    }
  Frags<BF16, 1> F1;
  make_frags<BF16, 1>(F1, g0);
  f32x4 c[NT];
  zero_tiles<NT>(c);
  panel_gemm<1, NT, BF16, T::ST_O2>(c, F1, P, ring, (unsigned)A.L.gto4, (unsigned)A.L.gto2, true, lane);          // d g4
  {
    f32x4 pre[NT];
    load_rows<NT>(pre, act + 2 * HD + D, quad);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[t][i] *= gelu_grad_f(pre[t][i]);                                            // d a_O2
  }
  Frags<BF16, KB> FD;
  make_frags<BF16, KB>(FD, c);
  zero_tiles<NT>(c);
  panel_gemm<KB, NT, BF16, T::ST_G0>(c, FD, P, ring, (unsigned)A.L.gto2, (unsigned)A.L.gto0, true, lane);         // d g3
  {
    f32x4 pre[NT];
    load_rows<NT>(pre, act + 2 * HD, quad);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[t][i] *= gelu_grad_f(pre[t][i]);                                            // d a_O0
  }
  make_frags<BF16, KB>(FD, c);
  f32x4 a[NTH];
  zero_tiles<NTH>(a);
  panel_gemm<KB, NTH, BF16, T::ST_TB>(a, FD, P, ring, (unsigned)A.L.gto0, (unsigned)A.L.gtf1, true, lane);        // d g2
  {
    f32x4 pre[NTH];
    load_rows<NTH>(pre, act + HD, quad);
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[t][i] *= gelu_grad_f(pre[t][i]);                                            // d a_F1
  }
  Frags<BF16, KBH> FH;
  make_frags<BF16, KBH>(FH, a);
  zero_tiles<NTH>(a);
  panel_gemm<KBH, NTH, BF16, T::ST_TB>(a, FH, P, ring, (unsigned)A.L.gtf1, (unsigned)A.L.gtb, true, lane);        // d n
  {
    // LayerNorm backward + gelu backward on a_B
    f32x4 pre[NTH];
    load_rows<NTH>(pre, act, quad);
    const float mu = act[T::ACT - 2], rstd = act[T::ACT - 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nh = (gelu_f(pre[t][i]) - mu) * rstd;
        s1 += a[t][i]; s2 = fmaf(a[t][i], nh, s2);
      }
    const float m1 = xquad_sum(s1) * A.inv_hd, m2 = xquad_sum(s2) * A.inv_hd;
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nh = (gelu_f(pre[t][i]) - mu) * rstd;
        a[t][i] = rstd * (a[t][i] - m1 - nh * m2) * gelu_grad_f(pre[t][i]);                                     // d a_B
      }
  }
  make_frags<BF16, KBH>(FH, a);
  zero_tiles<NTH>(a);
  panel_gemm<KBH, NTH, BF16, 1024>(a, FH, P, ring, (unsigned)A.L.gtb, NO_STAGE, true, lane);                     // d ybar
  f32x4 y[NTH];
  load_rows<NTH>(y, yrow, quad);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) s = fmaf(a[h * NT + t][i], y[h * NT + t][i], s);
    s = xquad_sum(s);
    if (quad == 0 && qvalid) A.delta[(size_t)qi * H + h] = s;
  }
  if (qvalid) store_rows<NTH>(a, A.dybar + (size_t)qi * HD, quad);
}

template <int D, int H, bool BF16>
static int launch_tail(const TailArgs& A, bool bwd, hipStream_t st) {
  using T = TailCfg<D, H, BF16>;
  static bool attr_set[2] = {false, false};
  const void* kern = bwd ? reinterpret_cast<const void*>(enf_tail_bwd_kernel<D, H, BF16>)
                         : reinterpret_cast<const void*>(enf_tail_fwd_kernel<D, H, BF16>);
  if (!attr_set[bwd]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM) != hipSuccess) return ENF_ELAUNCH;
    attr_set[bwd] = true;
  }
  dim3 grid((A.NQ + 16 * NWAVES - 1) / (16 * NWAVES));
  if (bwd) hipLaunchKernelGGL((enf_tail_bwd_kernel<D, H, BF16>), grid, dim3(NTHREADS), T::SMEM, st, A);
  else hipLaunchKernelGGL((enf_tail_fwd_kernel<D, H, BF16>), grid, dim3(NTHREADS), T::SMEM, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

extern "C" int enf_launch_tail(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, float* out,
                               const float* dout, float* dybar, float* delta, float* act, int bwd, hipStream_t st) {
  if (m.OB != 1) return ENF_EUNSUPPORTED;
  TailArgs A;
  A.ybar = ybar; A.blob = blob; A.L = L; A.out = out; A.dout = dout; A.dybar = dybar; A.delta = delta; A.act = act;
  A.NQ = m.B * m.N; A.O = m.O; A.inv_hd = 1.0f / (float)(m.Ht * m.Dt);
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) return m.bf16 ? launch_tail<DD, HH, true>(A, bwd != 0, st) : launch_tail<DD, HH, false>(A, bwd != 0, st);
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
  ENF_CASE(64, 4)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
