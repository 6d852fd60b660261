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
#include <cstdlib>
#include "enf_layout.h"
#include "enf_launch.h"
#define ENF_PIPE_ONE_RS 1      // the tail streams from the weight blob only (enf_device.h: stage_issue_p)
#include "enf_device.h"

struct TailArgs {
  const float* ybar; const char* blob; EnfLayout L;
  float* out;            // forward: (B*N, O)
  const float* dout;     // backward
  float* dybar; float* delta; float* act;   // backward outputs + scratch (B*N x (2HD + 2D + 2))
  const float* target; float* loss; float gscale, inv_n;     // fused loss (LOSS): d out = 2 (out - target) inv_n gscale, *loss += mean sq. error
  float* tdel;           // weight-gradient backward (WG): per query d a_B | d a_F1 | d a_O0 | d a_O2 (2HD + 2D floats); the layer INPUTS
                         // n^ | gelu(a_F1) | gelu(a_O0) | gelu(a_O2) replace the pre-activations in `act` (enf_train.hip forms X^T delta)
  int ybar_half;         // forward only (ENF_STAGE_YBAR_HALF): `ybar` holds bf16 rows
  int NQ, O;             // NQ = B*N queries
  float inv_hd;          // 1 / (H * true num_hidden)
};

// row-per-query global <-> acc layout (feature f = 16 t + 4 quad + i lives in tile t register i)
template <int NT> DEV void load_rows(f32x4 (&X)[NT], const float* row, int quad) {
#pragma unroll
  for (int t = 0; t < NT; ++t) X[t] = *reinterpret_cast<const f32x4*>(row + 16 * t + 4 * quad);
}
// the same from bf16 rows (two values per word, low half first)
template <int NT> DEV void load_rows_half(f32x4 (&X)[NT], const unsigned short* row, int quad) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const uint2 w = *reinterpret_cast<const uint2*>(row + 16 * t + 4 * quad);
    X[t] = f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                 __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
  }
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
  static constexpr int SMEM3 = 3 * STAGE_MAX + 4 * (2 * HD + 2 * D + 32);      // LA2: three ring slots
};

// ---- LA2: a 3-slot ring with TWO stages in flight.  The tail's GEMM stages are short (0.25 us of MFMAs per 32 KB
// stage) against ~1.1 us from LDS-DMA issue to landing, so with one stage in flight (panel_gemm) a small grid -- the fit
// shape has 64 workgroups, one per CU -- spends its time waiting for weights.  Here stage s + 2 of the kernel's stage
// stream is issued at the start of stage s; the end-of-stage wait lets that newest stage stay outstanding
// (s_waitcnt vmcnt(its per-wave instruction count)) and only requires stage s + 1.  One barrier per stage as before:
// a wave writes slot (s + 2) % 3 = (s - 1) % 3 only after every wave has left stage s - 1.
// (Large grids keep the 2-slot kernel: two workgroups per CU hide the latency there and need the LDS.)
template <int BYTES> struct La2Cnt { static constexpr int K = BYTES > 0 ? (BYTES / 1024) / NWAVES : 0; };
template <int K> DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }
template <int BYTES> DEV void la2_issue(const Pipe& P, unsigned off, char* ring, int slot, int lane) {
  stage_issue<BYTES>(P.rs, off, ring + slot * STAGE_MAX, P.wave, lane);
}
// panel descriptor for the stream: offset at run time, shape at compile time
template <int KBIN, int MTOUT, bool BF16> struct Pan {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  static constexpr int SPP = C::SPP, STAGE = C::STAGE;
};
struct NoPan { static constexpr int SPP = 0, STAGE = 0; };

// One stage step: issue the stage two ahead (own panel, else next panel N1, else the one after, N2), multiply, wait, barrier.
template <int SP, int KBIN, int MTOUT, bool BF16, typename N1, typename N2, int INIT>
DEV void la2_step(f32x4 (&acc)[MTOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel, unsigned n1, unsigned n2,
                  int lane, const float* bias) {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  constexpr int AHEAD = SP + 2 - C::SPP;          // < 0: own panel; 0.. : stage AHEAD of the following panels
  const int slot2 = (P.cur + 2) % 3;
  constexpr int ISSUED = AHEAD < 0 ? C::STAGE : (AHEAD < N1::SPP ? N1::STAGE : (AHEAD == N1::SPP && N1::SPP <= 1 ? N2::STAGE : 0));
  if constexpr (AHEAD < 0) la2_issue<C::STAGE>(P, panel + (SP + 2) * C::STAGE, ring, slot2, lane);
  else if constexpr (AHEAD < N1::SPP) la2_issue<N1::STAGE>(P, n1 + AHEAD * N1::STAGE, ring, slot2, lane);
  else if constexpr (AHEAD == N1::SPP && N1::SPP <= 1 && N2::STAGE > 0) la2_issue<N2::STAGE>(P, n2, ring, slot2, lane);
  gemm_stage<BF16, KBIN, C::MTS, INIT>(&acc[SP * C::MTS], F, ring + P.cur * STAGE_MAX, lane, bias + 16 * SP * C::MTS);
  wait_vm<La2Cnt<ISSUED>::K>();
  __syncthreads();
  P.cur = (P.cur + 1) % 3;
}
template <int KBIN, int MTOUT, bool BF16, typename N1, typename N2, int INIT = INIT_ACC>
DEV void panel_gemm_la2(f32x4 (&acc)[MTOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel, unsigned n1,
                        unsigned n2, int lane, const float* bias = nullptr) {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  static_assert(C::SPP <= 8, "unrolled by hand");
#define LA2_STEP(SP_) if constexpr (C::SPP > SP_) la2_step<SP_, KBIN, MTOUT, BF16, N1, N2, INIT>(acc, F, P, ring, panel, n1, n2, lane, bias);
  LA2_STEP(0) LA2_STEP(1) LA2_STEP(2) LA2_STEP(3) LA2_STEP(4) LA2_STEP(5) LA2_STEP(6) LA2_STEP(7)
#undef LA2_STEP
}
// stream start: stage 0 and stage 1 (of the first panel A, or of the second panel B when A has one stage)
template <typename A0, typename B0>
DEV void la2_first(Pipe& P, char* ring, unsigned a, unsigned b, int wave, int lane) {
  P.cur = 0;
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  P.early = false;
  la2_issue<A0::STAGE>(P, a, ring, 0, lane);
  constexpr int S1 = A0::SPP > 1 ? A0::STAGE : B0::STAGE;
  if constexpr (A0::SPP > 1) la2_issue<A0::STAGE>(P, a + A0::STAGE, ring, 1, lane);
  else if constexpr (B0::STAGE > 0) la2_issue<B0::STAGE>(P, b, ring, 1, lane);
  wait_vm<La2Cnt<S1>::K>();
  __syncthreads();
}
// dispatch: LA2 or the 2-slot panel_gemm
template <bool LA2, int KBIN, int MTOUT, bool BF16, typename N1, typename N2, int NEXT_BYTES>
DEV void tail_gemm(f32x4 (&acc)[MTOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel, unsigned n1, unsigned n2,
                   int lane) {
  if constexpr (LA2) panel_gemm_la2<KBIN, MTOUT, BF16, N1, N2>(acc, F, P, ring, panel, n1, n2, lane);
  else panel_gemm<KBIN, MTOUT, BF16, NEXT_BYTES>(acc, F, P, ring, panel, n1, true, lane);
}

// Forward chain for this wave's 16 queries.  o4 = the (padded) 32 network outputs.
// SAVE: stash pre-activations + LN stats to `act` (row per query) for the backward chain.
// NX1 / NX2 (Pan<..> or NoPan) + next / next2: the two panels that follow the forward chain (the backward chain's first
// two, or nothing); the 2-slot pipeline only uses the first stage of NX1.
template <int D, int H, bool BF16, bool SAVE, bool LA2, typename NX1, typename NX2>
DEV void tail_forward(f32x4 (&o4)[2], const float* yrow, float* act, const EnfLayout& L, const float* cst, Pipe& P,
                      char* ring, unsigned next, unsigned next2, int lane, int quad, float inv_hd, bool save_ok = true, bool y_half = false) {
  using T = TailCfg<D, H, BF16>;
  constexpr int KB = T::KB, KBH = T::KBH, NT = T::NT, NTH = T::NTH, HD = T::HD;
  using PTB = Pan<KBH, NTH, BF16>; using PO0 = Pan<KBH, NT, BF16>; using PO2 = Pan<KB, NT, BF16>; using PO4 = Pan<KB, 2, BF16>;
  constexpr int NEXT_BYTES = NX1::STAGE > 0 ? NX1::STAGE : 1024;
  const float* c_bB = cst, *c_bF1 = cst + HD, *c_bO0 = cst + 2 * HD, *c_bO2 = cst + 2 * HD + D, *c_bO4 = cst + 2 * HD + 2 * D;
  Frags<BF16, KBH> FH;
  f32x4 a[NTH];
  if (y_half) load_rows_half<NTH>(a, reinterpret_cast<const unsigned short*>(yrow), quad);     // (yrow: the caller's row pointer in bf16 units)
  else load_rows<NTH>(a, yrow, quad);
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int t = 0; t < NTH; ++t) a[t] = rowvec(c_bB, t, quad);
  tail_gemm<LA2, KBH, NTH, BF16, PTB, PO0, T::ST_TB>(a, FH, P, ring, (unsigned)L.atb, (unsigned)L.atf1, (unsigned)L.ato0, lane);
  if (SAVE && save_ok) store_rows<NTH>(a, act, quad);
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[t][i] = gelu_f(a[t][i]);
  float mu, rstd;
  ln_stats<NTH>(a, mu, rstd, inv_hd);
  if (SAVE && save_ok && quad == 0) { act[T::ACT - 2] = mu; act[T::ACT - 1] = rstd; }
  ln_apply<NTH>(a, mu, rstd);          // (scalar fmas on purpose: enf_device.h)
  make_frags<BF16, KBH>(FH, a);
#pragma unroll
  for (int t = 0; t < NTH; ++t) a[t] = rowvec(c_bF1, t, quad);
  tail_gemm<LA2, KBH, NTH, BF16, PO0, PO2, T::ST_O0>(a, FH, P, ring, (unsigned)L.atf1, (unsigned)L.ato0, (unsigned)L.ato2, lane);
  if (SAVE && save_ok) store_rows<NTH>(a, act + HD, quad);
#pragma unroll
  for (int t = 0; t < NTH; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[t][i] = gelu_f(a[t][i]);       // NEF:230
  make_frags<BF16, KBH>(FH, a);
  f32x4 c[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) c[t] = rowvec(c_bO0, t, quad);
  tail_gemm<LA2, KBH, NT, BF16, PO2, PO4, T::ST_O2>(c, FH, P, ring, (unsigned)L.ato0, (unsigned)L.ato2, (unsigned)L.ato4, lane);
  if (SAVE && save_ok) store_rows<NT>(c, act + 2 * HD, quad);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) c[t][i] = gelu_f(c[t][i]);
  Frags<BF16, KB> FD;
  make_frags<BF16, KB>(FD, c);
#pragma unroll
  for (int t = 0; t < NT; ++t) c[t] = rowvec(c_bO2, t, quad);
  tail_gemm<LA2, KB, NT, BF16, PO4, NX1, T::ST_O4>(c, FD, P, ring, (unsigned)L.ato2, (unsigned)L.ato4, next, lane);
  if (SAVE && save_ok) store_rows<NT>(c, act + 2 * HD + D, quad);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) c[t][i] = gelu_f(c[t][i]);
  make_frags<BF16, KB>(FD, c);
  o4[0] = rowvec(c_bO4, 0, quad);
  o4[1] = rowvec(c_bO4, 1, quad);
  tail_gemm<LA2, KB, 2, BF16, NX1, NX2, NEXT_BYTES>(o4, FD, P, ring, (unsigned)L.ato4, next, next2, lane);
}

template <int D, int H, bool BF16>
DEV void tail_consts(float* cst, const char* blob, const EnfLayout& L, int tid) {
  constexpr int HD = H * D;
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };
  for (int i = tid; i < HD; i += NTHREADS) { cst[i] = G(L.bB)[i]; cst[HD + i] = G(L.bF1)[i]; }
  for (int i = tid; i < D; i += NTHREADS) { cst[2 * HD + i] = G(L.bO0)[i]; cst[2 * HD + D + i] = G(L.bO2)[i]; }
  for (int i = tid; i < 32; i += NTHREADS) cst[2 * HD + 2 * D + i] = G(L.bO4)[i];
}

template <int D, int H, bool BF16, bool LA2, bool SAVE>
__global__ __launch_bounds__(NTHREADS, 2) void enf_tail_fwd_kernel(TailArgs A) {
  using T = TailCfg<D, H, BF16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + (LA2 ? 3 : 2) * STAGE_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, quad = lane >> 4;
  const int q0 = blockIdx.x * (16 * NWAVES) + wave * 16;
  const int qi = min(q0 + col, A.NQ - 1);
  tail_consts<D, H, BF16>(cst, A.blob, A.L, tid);
  Pipe P;
  P.rs = make_blob_rsrc(A.blob, (unsigned)A.L.total);
  P.rs2 = P.rs;
  if constexpr (LA2) {
    __syncthreads();                                   // the LDS constants, for waves that run ahead of the first stage barrier
    la2_first<Pan<T::KBH, T::NTH, BF16>, Pan<T::KBH, T::NTH, BF16>>(P, ring, (unsigned)A.L.atb, (unsigned)A.L.atf1, wave, lane);
  } else first_stage<T::ST_TB>(P, ring, (unsigned)A.L.atb, wave, lane);
  f32x4 o4[2];
  // SAVE: the pre-activations go to the workspace so that the backward that follows need not recompute this chain
  const bool yh = !SAVE && A.ybar_half;
  const float* yrow = yh ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(A.ybar) + (size_t)qi * T::HD)
                         : A.ybar + (size_t)qi * T::HD;
  tail_forward<D, H, BF16, SAVE, LA2, NoPan, NoPan>(o4, yrow, SAVE ? A.act + (size_t)qi * T::ACT : nullptr, A.L,
                                                    cst, P, ring, NO_STAGE, NO_STAGE, lane, quad, A.inv_hd, true, yh);
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

// LOSS (with RECOMP): the inner step's tail in ONE kernel -- forward chain, the reconstruction loss and its gradient
// (pde_trainer.py:185: mean squared error over all B N O outputs; enf_loss.hip's arithmetic) formed in registers from the forward's
// outputs, backward chain: no `out` / `d out` round trip and two launches less between the pair kernels of an inner step.
template <int D, int H, bool BF16, bool LA2, bool RECOMP, bool WG = false, bool LOSS = false>
__global__ __launch_bounds__(NTHREADS, 2) void enf_tail_bwd_kernel(TailArgs A) {
  static_assert(!LOSS || RECOMP, "the fused loss needs the forward chain's outputs");
  using T = TailCfg<D, H, BF16>;
  constexpr int KB = T::KB, KBH = T::KBH, NT = T::NT, NTH = T::NTH, HD = T::HD;
  using PG4 = Pan<1, NT, BF16>; using PG2 = Pan<KB, NT, BF16>; using PG0 = Pan<KB, NTH, BF16>; using PGH = Pan<KBH, NTH, BF16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  float* cst = reinterpret_cast<float*>(smem + (LA2 ? 3 : 2) * STAGE_MAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, quad = lane >> 4;
  const int q0 = blockIdx.x * (16 * NWAVES) + wave * 16;
  const bool qvalid = q0 + col < A.NQ;
  const int qi = min(q0 + col, A.NQ - 1);
  // clamped (duplicate) queries write the same scratch values: benign
  float* act = A.act + (size_t)qi * T::ACT;
  float* tdel = WG ? A.tdel + (size_t)qi * (2 * HD + 2 * D) : nullptr;
  const float* yrow = A.ybar + (size_t)qi * HD;
  tail_consts<D, H, BF16>(cst, A.blob, A.L, tid);
  Pipe P;
  P.rs = make_blob_rsrc(A.blob, (unsigned)A.L.total);
  P.rs2 = P.rs;
  f32x4 o4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if constexpr (RECOMP) {
    if constexpr (LA2) {
      __syncthreads();
      la2_first<PGH, PGH>(P, ring, (unsigned)A.L.atb, (unsigned)A.L.atf1, wave, lane);
    } else first_stage<T::ST_TB>(P, ring, (unsigned)A.L.atb, wave, lane);
    // WG: the rows of `act` are rewritten in place below (layer inputs over pre-activations), so a clamped duplicate lane -- it shares
    // its row with the query's own lane in another wave -- must not store: its late pre-activation would land on the finished row
    tail_forward<D, H, BF16, true, LA2, PG4, PG2>(o4, yrow, act, A.L, cst, P, ring, (unsigned)A.L.gto4, (unsigned)A.L.gto2, lane, quad,
                                                  A.inv_hd, !WG || qvalid);
  } else {      // the forward of this step stashed the pre-activations (enf_tail_fwd_kernel<.., SAVE>): start at the backward chain
    if constexpr (LA2) {
      __syncthreads();
      la2_first<PG4, PG2>(P, ring, (unsigned)A.L.gto4, (unsigned)A.L.gto2, wave, lane);
    } else first_stage<T::ST_G4>(P, ring, (unsigned)A.L.gto4, wave, lane);
  }
  // the pre-activations this lane stored are re-read by this lane only (same addresses)

  // ---- backward chain
  f32x4 g0[2];
  if constexpr (LOSS) {
    float se = 0.f;
    const float gs = 2.0f * A.inv_n * A.gscale;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = 16 * t + 4 * quad + i;
        const float dd = (o < A.O && qvalid) ? o4[t][i] - A.target[(size_t)qi * A.O + o] : 0.f;
        se = fmaf(dd, dd, se);
        g0[t][i] = dd * gs;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    if (lane == 0) atomicAdd(A.loss, se * A.inv_n);          // (one per wave, nobody waits for it; the caller zeroed *loss)
  } else {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = 16 * t + 4 * quad + i;
      g0[t][i] = (o < A.O && qvalid) ? A.dout[(size_t)qi * A.O + o] : 0.f;
    }
  }
  Frags<BF16, 1> F1;
  make_frags<BF16, 1>(F1, g0);
  f32x4 c[NT];
  zero_tiles<NT>(c);
  tail_gemm<LA2, 1, NT, BF16, PG2, PG0, T::ST_O2>(c, F1, P, ring, (unsigned)A.L.gto4, (unsigned)A.L.gto2, (unsigned)A.L.gto0, lane);   // d g4
  {
    f32x4 pre[NT];
    load_rows<NT>(pre, act + 2 * HD + D, quad);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[t][i] *= gelu_grad_f(pre[t][i]);                                            // d a_O2
    if constexpr (WG) {
      if (qvalid) store_rows<NT>(c, tdel + 2 * HD + D, quad);      // (a clamped duplicate lane holds zeros: only the query's own lane writes)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[t][i] = gelu_f(pre[t][i]);
      if (qvalid) store_rows<NT>(pre, act + 2 * HD + D, quad);                                                  // input of out_proj.layers_4
    }
  }
  Frags<BF16, KB> FD;
  make_frags<BF16, KB>(FD, c);
  zero_tiles<NT>(c);
  tail_gemm<LA2, KB, NT, BF16, PG0, PGH, T::ST_G0>(c, FD, P, ring, (unsigned)A.L.gto2, (unsigned)A.L.gto0, (unsigned)A.L.gtf1, lane);  // d g3
  {
    f32x4 pre[NT];
    load_rows<NT>(pre, act + 2 * HD, quad);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[t][i] *= gelu_grad_f(pre[t][i]);                                            // d a_O0
    if constexpr (WG) {
      if (qvalid) store_rows<NT>(c, tdel + 2 * HD, quad);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[t][i] = gelu_f(pre[t][i]);
      if (qvalid) store_rows<NT>(pre, act + 2 * HD, quad);                                                      // input of out_proj.layers_2
    }
  }
  make_frags<BF16, KB>(FD, c);
  f32x4 a[NTH];
  zero_tiles<NTH>(a);
  tail_gemm<LA2, KB, NTH, BF16, PGH, PGH, T::ST_TB>(a, FD, P, ring, (unsigned)A.L.gto0, (unsigned)A.L.gtf1, (unsigned)A.L.gtb, lane);  // d g2
  {
    f32x4 pre[NTH];
    load_rows<NTH>(pre, act + HD, quad);
#pragma unroll
    for (int t = 0; t < NTH; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[t][i] *= gelu_grad_f(pre[t][i]);                                            // d a_F1
    if constexpr (WG) {
      if (qvalid) store_rows<NTH>(a, tdel + HD, quad);
#pragma unroll
      for (int t = 0; t < NTH; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[t][i] = gelu_f(pre[t][i]);
      if (qvalid) store_rows<NTH>(pre, act + HD, quad);                                                         // input of out_proj.layers_0
    }
  }
  Frags<BF16, KBH> FH;
  make_frags<BF16, KBH>(FH, a);
  zero_tiles<NTH>(a);
  tail_gemm<LA2, KBH, NTH, BF16, PGH, NoPan, T::ST_TB>(a, FH, P, ring, (unsigned)A.L.gtf1, (unsigned)A.L.gtb, NO_STAGE, lane);          // d n
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
        if constexpr (WG) pre[t][i] = nh;
      }
    if constexpr (WG) {
      if (qvalid) store_rows<NTH>(a, tdel, quad);
      if (qvalid) store_rows<NTH>(pre, act, quad);                                                              // n^: input of the (folded) FFN Dense_1
    }
  }
  make_frags<BF16, KBH>(FH, a);
  zero_tiles<NTH>(a);
  tail_gemm<LA2, KBH, NTH, BF16, NoPan, NoPan, 1024>(a, FH, P, ring, (unsigned)A.L.gtb, NO_STAGE, NO_STAGE, lane);                       // d ybar
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
static int launch_tail(const TailArgs& A, bool bwd, bool opt, hipStream_t st) {
  const bool wg = bwd && A.tdel != nullptr;
  const bool fused_loss = bwd && A.target != nullptr;
  using T = TailCfg<D, H, BF16>;
  dim3 grid((A.NQ + 16 * NWAVES - 1) / (16 * NWAVES));
  // few workgroups (at most one per CU): the deeper weight pipeline (LA2) instead of a second workgroup per CU
#ifdef ENF_AB_SWITCHES       // A/B builds only: ENF_TAIL_LA2=0/1 forces the choice
  static int la2_mode = -1;
  if (la2_mode < 0) { const char* e = getenv("ENF_TAIL_LA2"); la2_mode = e ? (e[0] == '0' ? 0 : 1) : 2; }
  const bool la2 = la2_mode == 2 ? grid.x <= 256 : la2_mode == 1;
#else
  const bool la2 = grid.x <= 256;
#endif
  // opt: forward -> stash the pre-activations (SAVE); backward -> they are stashed, skip the recompute
  static EnfAttrBits attr_done[4][2][2];     // [fwd / bwd / bwd + WG][la2][opt] (this function is one instantiation per D, H, BF16), one bit per device
  auto go = [&](void (*kern_ptr)(TailArgs)) -> int {
    const int smem = la2 ? T::SMEM3 : T::SMEM;
    if (!enf_lds_attr(reinterpret_cast<const void*>(kern_ptr), smem, attr_done[fused_loss ? 3 : (wg ? 2 : bwd)][la2][opt])) return ENF_ELAUNCH;
    hipLaunchKernelGGL(kern_ptr, grid, dim3(NTHREADS), smem, st, A);
    return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
  };
  if (fused_loss) {
    if (wg || opt) return ENF_EINVAL;
    return la2 ? go(enf_tail_bwd_kernel<D, H, BF16, true, true, false, true>) : go(enf_tail_bwd_kernel<D, H, BF16, false, true, false, true>);
  }
  if (wg) {
    if (la2) return opt ? go(enf_tail_bwd_kernel<D, H, BF16, true, false, true>) : go(enf_tail_bwd_kernel<D, H, BF16, true, true, true>);
    return opt ? go(enf_tail_bwd_kernel<D, H, BF16, false, false, true>) : go(enf_tail_bwd_kernel<D, H, BF16, false, true, true>);
  }
  if (bwd) {
    if (la2) return opt ? go(enf_tail_bwd_kernel<D, H, BF16, true, false>) : go(enf_tail_bwd_kernel<D, H, BF16, true, true>);
    return opt ? go(enf_tail_bwd_kernel<D, H, BF16, false, false>) : go(enf_tail_bwd_kernel<D, H, BF16, false, true>);
  }
  if (la2) return opt ? go(enf_tail_fwd_kernel<D, H, BF16, true, true>) : go(enf_tail_fwd_kernel<D, H, BF16, true, false>);
  return opt ? go(enf_tail_fwd_kernel<D, H, BF16, false, true>) : go(enf_tail_fwd_kernel<D, H, BF16, false, false>);
}

extern "C" int enf_launch_tail_wg(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, float* out,
                                  const float* dout, float* dybar, float* delta, float* act, float* tdel, int bwd, int opt, hipStream_t st);
extern "C" int enf_launch_tail(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, float* out,
                               const float* dout, float* dybar, float* delta, float* act, int bwd, int opt, hipStream_t st) {
  return enf_launch_tail_wg(m, L, blob, ybar, out, dout, dybar, delta, act, nullptr, bwd, opt, st);
}
// tdel != NULL (backward only): the weight-gradient form of the backward -- it also leaves every layer's input (in `act`, in place
// of the pre-activations) and delta (in `tdel`) for the X^T delta products of enf_train.hip
extern "C" int enf_launch_tail_loss(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, const float* target,
                                    float gscale, float* loss, float* dybar, float* delta, float* act, hipStream_t st);
extern "C" int enf_launch_tail_wg(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, float* out,
                                  const float* dout, float* dybar, float* delta, float* act, float* tdel, int bwd, int opt, hipStream_t st) {
  if (m.OB != 1) return ENF_EUNSUPPORTED;
  TailArgs A;
  A.target = nullptr; A.loss = nullptr; A.gscale = 0.f; A.inv_n = 0.f;
  A.tdel = tdel;
  A.ybar_half = (!bwd && (opt & 2) && m.bf16) ? 1 : 0;
  opt &= 1;
  A.ybar = ybar; A.blob = blob; A.L = L; A.out = out; A.dout = dout; A.dybar = dybar; A.delta = delta; A.act = act;
  A.NQ = m.B * m.N; A.O = m.O; A.inv_hd = 1.0f / (float)(m.Ht * m.Dt);
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) return m.bf16 ? launch_tail<DD, HH, true>(A, bwd != 0, opt != 0, st) : launch_tail<DD, HH, false>(A, bwd != 0, opt != 0, st);
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
  ENF_CASE(64, 4)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}

// the inner step's tail as one kernel: forward chain -> mean squared error against `target` (added to *loss) and its gradient ->
// backward chain -> d ybar, delta
extern "C" int enf_launch_tail_loss(const EnfDims& m, const EnfLayout& L, const char* blob, const float* ybar, const float* target,
                                    float gscale, float* loss, float* dybar, float* delta, float* act, hipStream_t st) {
  if (m.OB != 1) return ENF_EUNSUPPORTED;
  TailArgs A;
  A.ybar = ybar; A.blob = blob; A.L = L; A.out = nullptr; A.dout = nullptr; A.dybar = dybar; A.delta = delta; A.act = act; A.tdel = nullptr;
  A.ybar_half = 0;
  A.target = target; A.loss = loss; A.gscale = gscale; A.inv_n = 1.0f / ((float)m.B * (float)m.N * (float)m.O);
  A.NQ = m.B * m.N; A.O = m.O; A.inv_hd = 1.0f / (float)(m.Ht * m.Dt);
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) return m.bf16 ? launch_tail<DD, HH, true>(A, true, false, st) : launch_tail<DD, HH, false>(A, true, false, st);
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
  ENF_CASE(64, 4)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
