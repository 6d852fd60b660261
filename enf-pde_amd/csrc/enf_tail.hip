// enf_tail.hip -- per-query tail of the decoder, forward and backward.
//
// Input: ybar (B,N,H*D) = softmax-weighted sum of the mixer's normalised gelu output (K2).
// Because softmax weights sum to 1, everything linear after ECA:144 is applied once per query:
//   WB  = blockdiag(diag(LN.scale) Dense_1 [mixer, ECA:19-20]) . attn.out_proj (ECA:150) . ffn.Dense_0 (ECA:17)
//   gelu, LayerNorm (ECA:18-19), WF1 = diag(LN.scale) ffn.Dense_1 (ECA:20)            [block FFN, NEF:66]
//   gelu (NEF:230), out_proj: Dense, gelu, Dense, gelu, Dense (NEF:196-202,233)
// One wave = 32 queries, transposed chain as in enf_device.h; 4 waves share the LDS weight ring.
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
};

// row-per-query global <-> acc layout (feature f = 32k + 8g + 4half + i lives in reg 4g+i)
template <int NB> DEV void load_rows(f32x16 (&X)[NB], const float* row, int half) {
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * k + 8 * g + 4 * half);
      X[k][4 * g] = v[0]; X[k][4 * g + 1] = v[1]; X[k][4 * g + 2] = v[2]; X[k][4 * g + 3] = v[3];
    }
}
template <int NB> DEV void store_rows(const f32x16 (&X)[NB], float* row, int half) {
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
      v[0] = X[k][4 * g]; v[1] = X[k][4 * g + 1]; v[2] = X[k][4 * g + 2]; v[3] = X[k][4 * g + 3];
      *reinterpret_cast<f32x4*>(row + 32 * k + 8 * g + 4 * half) = v;
    }
}

template <int D, int H, bool BF16> struct TailCfg {
  static constexpr int KB = D / 32, KBH = H * D / 32, HD = H * D;
  static constexpr int ACT = 2 * HD + 2 * D + 2;   // a_B | a_F1 | a_O0 | a_O2 | mu | rstd
  static constexpr int ST_TB = PanelCfg<KBH, KBH, BF16>::STAGE;
  static constexpr int ST_O0 = PanelCfg<KBH, KB, BF16>::STAGE;
  static constexpr int ST_O2 = PanelCfg<KB, KB, BF16>::STAGE;
  static constexpr int ST_O4 = PanelCfg<KB, 1, BF16>::STAGE;    // per out-block of the last layer
  static constexpr int ST_G4 = PanelCfg<1, KB, BF16>::STAGE;    // gto4: in = 1 block (OB == 1)
  static constexpr int ST_G0 = PanelCfg<KB, KBH, BF16>::STAGE;  // gto0: rows HD, K = D
  static constexpr int SMEM = 2 * STAGE_MAX + 4 * (2 * HD + 2 * D + 32);
};

// Forward chain for this wave's 32 queries.  Y: in = ybar, out (first block) = network output.
// SAVE: stash pre-activations + LN stats to `act` (row per query) for the backward chain.
template <int D, int H, bool BF16, bool SAVE, int NEXT_BYTES>
DEV void tail_forward(f32x16 (&o4)[1], const float* yrow, float* act, const char* blob, const EnfLayout& L,
                      const float* cst, Pipe& P, char* ring, unsigned next, int tid, int lane, int half) {
  using T = TailCfg<D, H, BF16>;
  constexpr int KB = T::KB, KBH = T::KBH, HD = T::HD;
  const float* c_bB = cst, *c_bF1 = cst + HD, *c_bO0 = cst + 2 * HD, *c_bO2 = cst + 2 * HD + D, *c_bO4 = cst + 2 * HD + 2 * D;
  Frags<BF16, KBH> FH;
  f32x16 a[KBH];
  load_rows<KBH>(a, yrow, half);
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int k = 0; k < KBH; ++k) load_rowvec(a[k], c_bB, k, half);
  panel_gemm<KBH, KBH, BF16, T::ST_TB>(a, FH, P, ring, (unsigned)L.atb, (unsigned)L.atf1, true, tid, lane);
  if (SAVE) store_rows<KBH>(a, act, half);
#pragma unroll
  for (int k = 0; k < KBH; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[k][r] = gelu_f(a[k][r]);
  float mu, rstd;
  ln_stats<KBH>(a, mu, rstd);
  if (SAVE && half == 0) { act[T::ACT - 2] = mu; act[T::ACT - 1] = rstd; }
#pragma unroll
  for (int k = 0; k < KBH; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[k][r] = (a[k][r] - mu) * rstd;
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int k = 0; k < KBH; ++k) load_rowvec(a[k], c_bF1, k, half);
  panel_gemm<KBH, KBH, BF16, T::ST_O0>(a, FH, P, ring, (unsigned)L.atf1, (unsigned)L.ato0, true, tid, lane);
  if (SAVE) store_rows<KBH>(a, act + HD, half);
#pragma unroll
  for (int k = 0; k < KBH; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[k][r] = gelu_f(a[k][r]);       // NEF:230
  make_frags<BF16, KBH>(FH, a);
  f32x16 c[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) load_rowvec(c[k], c_bO0, k, half);
  panel_gemm<KBH, KB, BF16, T::ST_O2>(c, FH, P, ring, (unsigned)L.ato0, (unsigned)L.ato2, true, tid, lane);
  if (SAVE) store_rows<KB>(c, act + 2 * HD, half);
#pragma unroll
  for (int k = 0; k < KB; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = gelu_f(c[k][r]);
  Frags<BF16, KB> FD;
  make_frags<BF16, KB>(FD, c);
#pragma unroll
  for (int k = 0; k < KB; ++k) load_rowvec(c[k], c_bO2, k, half);
  panel_gemm<KB, KB, BF16, T::ST_O4>(c, FD, P, ring, (unsigned)L.ato2, (unsigned)L.ato4, true, tid, lane);
  if (SAVE) store_rows<KB>(c, act + 2 * HD + D, half);
#pragma unroll
  for (int k = 0; k < KB; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = gelu_f(c[k][r]);
  make_frags<BF16, KB>(FD, c);
  load_rowvec(o4[0], c_bO4, 0, half);
  panel_gemm<KB, 1, BF16, NEXT_BYTES>(o4, FD, P, ring, (unsigned)L.ato4, next, true, tid, lane);
}

template <int D, int H, bool BF16>
DEV void tail_consts(float* cst, const char* blob, const EnfLayout& L, int tid) {
  constexpr int HD = H * D;
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };
  for (int i = tid; i < HD; i += 256) { cst[i] = G(L.bB)[i]; cst[HD + i] = G(L.bF1)[i]; }
  for (int i = tid; i < D; i += 256) { cst[2 * HD + i] = G(L.bO0)[i]; cst[2 * HD + D + i] = G(L.bO2)[i]; }
  for (int i = tid; i < 32; i += 256) cst[2 * HD + 2 * D + i] = G(L.bO4)[i];
}

template <int D, int H, bool BF16>
__global__ __launch_bounds__(256, 1) void enf_tail_fwd_kernel(TailArgs A) {
  using T = TailCfg<D, H, BF16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + 2 * STAGE_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = min(q0 + col, A.NQ - 1);
  tail_consts<D, H, BF16>(cst, A.blob, A.L, tid);
  Pipe P;
  P.cur = 0;
  P.rs = make_blob_rsrc(A.blob, (unsigned)A.L.total);
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  stage_issue<T::ST_TB>(P.rs, (unsigned)A.L.atb, ring, P.wave, lane);
  stage_wait();
  __syncthreads();
  f32x16 o4[1];
  tail_forward<D, H, BF16, false, 4096>(o4, A.ybar + (size_t)qi * T::HD, nullptr, A.blob, A.L, cst, P, ring, NO_STAGE, tid, lane, half);
  if (q0 + col < A.NQ) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = RHO(r, 0) + 4 * half;
      if (o < A.O) A.out[(size_t)(q0 + col) * A.O + o] = o4[0][r];
    }
  }
}

template <int D, int H, bool BF16>
__global__ __launch_bounds__(256, 1) void enf_tail_bwd_kernel(TailArgs A) {
  using T = TailCfg<D, H, BF16>;
  constexpr int KB = T::KB, KBH = T::KBH, HD = T::HD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + 2 * STAGE_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool qvalid = q0 + col < A.NQ;
  const int qi = min(q0 + col, A.NQ - 1);
  // rows of clamped (duplicate) queries write the same scratch values: benign
  float* act = A.act + (size_t)qi * T::ACT;
  const float* yrow = A.ybar + (size_t)qi * HD;
  tail_consts<D, H, BF16>(cst, A.blob, A.L, tid);
  Pipe P;
  P.cur = 0;
  P.rs = make_blob_rsrc(A.blob, (unsigned)A.L.total);
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  stage_issue<T::ST_TB>(P.rs, (unsigned)A.L.atb, ring, P.wave, lane);
  stage_wait();
  __syncthreads();
  const char* blob = A.blob;
  f32x16 o4[1];
  tail_forward<D, H, BF16, true, T::ST_G4>(o4, yrow, act, blob, A.L, cst, P, ring, (unsigned)A.L.gto4, tid, lane, half);
  // the pre-activations this lane stored are re-read by this lane only (same addresses)

  // ---- backward chain
  f32x16 g0[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = RHO(r, 0) + 4 * half;
    g0[0][r] = (o < A.O && qvalid) ? A.dout[(size_t)qi * A.O + o] : 0.f;
  }
  Frags<BF16, 1> F1;
  make_frags<BF16, 1>(F1, g0);
  f32x16 c[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  panel_gemm<1, KB, BF16, T::ST_O2>(c, F1, P, ring, (unsigned)A.L.gto4, (unsigned)A.L.gto2, true, tid, lane);   // d g4
  {
    f32x16 pre[KB];
    load_rows<KB>(pre, act + 2 * HD + D, half);
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[k][r] *= gelu_grad_f(pre[k][r]);                                 // d a_O2
  }
  Frags<BF16, KB> FD;
  make_frags<BF16, KB>(FD, c);
#pragma unroll
  for (int k = 0; k < KB; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  panel_gemm<KB, KB, BF16, T::ST_G0>(c, FD, P, ring, (unsigned)A.L.gto2, (unsigned)A.L.gto0, true, tid, lane);  // d g3
  {
    f32x16 pre[KB];
    load_rows<KB>(pre, act + 2 * HD, half);
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[k][r] *= gelu_grad_f(pre[k][r]);                                 // d a_O0
  }
  make_frags<BF16, KB>(FD, c);
  f32x16 a[KBH];
#pragma unroll
  for (int k = 0; k < KBH; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[k][r] = 0.f;
  panel_gemm<KB, KBH, BF16, T::ST_TB>(a, FD, P, ring, (unsigned)A.L.gto0, (unsigned)A.L.gtf1, true, tid, lane); // d g2
  {
    f32x16 pre[KBH];
    load_rows<KBH>(pre, act + HD, half);
#pragma unroll
    for (int k = 0; k < KBH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[k][r] *= gelu_grad_f(pre[k][r]);                                 // d a_F1
  }
  Frags<BF16, KBH> FH;
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int k = 0; k < KBH; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[k][r] = 0.f;
  panel_gemm<KBH, KBH, BF16, T::ST_TB>(a, FH, P, ring, (unsigned)A.L.gtf1, (unsigned)A.L.gtb, true, tid, lane); // d n
  {
    // LayerNorm backward + gelu backward on a_B
    f32x16 pre[KBH];
    load_rows<KBH>(pre, act, half);
    const float mu = act[T::ACT - 2], rstd = act[T::ACT - 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KBH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float nh = (gelu_f(pre[k][r]) - mu) * rstd;
        s1 += a[k][r]; s2 = fmaf(a[k][r], nh, s2);
      }
    const float m1 = xhalf_sum(s1) * (1.0f / HD), m2 = xhalf_sum(s2) * (1.0f / HD);
#pragma unroll
    for (int k = 0; k < KBH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float nh = (gelu_f(pre[k][r]) - mu) * rstd;
        a[k][r] = rstd * (a[k][r] - m1 - nh * m2) * gelu_grad_f(pre[k][r]);                           // d a_B
      }
  }
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int k = 0; k < KBH; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[k][r] = 0.f;
  panel_gemm<KBH, KBH, BF16, 4096>(a, FH, P, ring, (unsigned)A.L.gtb, NO_STAGE, true, tid, lane);              // d ybar
  if (qvalid) {
    store_rows<KBH>(a, A.dybar + (size_t)qi * HD, half);
    f32x16 y[KBH];
    load_rows<KBH>(y, yrow, half);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(a[h * KB + k][r], y[h * KB + k][r], s);
      s = xhalf_sum(s);
      if (half == 0) A.delta[(size_t)qi * H + h] = s;
    }
  } else {
    // keep the shuffle inside xhalf_sum convergent for the whole wave
#pragma unroll
    for (int h = 0; h < H; ++h) (void)xhalf_sum(0.f);
  }
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
  dim3 grid((A.NQ + 127) / 128);
  if (bwd) hipLaunchKernelGGL((enf_tail_bwd_kernel<D, H, BF16>), grid, dim3(256), T::SMEM, st, A);
  else hipLaunchKernelGGL((enf_tail_fwd_kernel<D, H, BF16>), grid, dim3(256), T::SMEM, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

extern "C" int enf_launch_tail(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, float* out,
                               const float* dout, float* dybar, float* delta, float* act, int bwd, hipStream_t st) {
  if (m.OB != 1) return ENF_EUNSUPPORTED;
  TailArgs A;
  A.ybar = ybar; A.blob = blob; A.L = L; A.out = out; A.dout = dout; A.dybar = dybar; A.delta = delta; A.act = act;
  A.NQ = m.B * m.N; A.O = m.O;
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) return m.bf16 ? launch_tail<DD, HH, true>(A, bwd != 0, st) : launch_tail<DD, HH, false>(A, bwd != 0, st);
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
