// enf_pair_common.h -- pieces shared by the forward (K2) and backward (K3) pair kernels.
#pragma once
#include "enf_device.h"

template <int D, bool BF16> struct PairCfg {
  static constexpr int KB = D / 32;
  using DD = PanelCfg<KB, KB, BF16>;            // a D x D panel
  using GB = PanelCfg<KB, 2 * KB, BF16>;        // one head's gamma/beta panel (2D outputs)
  static_assert(GB::MBS % 2 == 0, "gamma/beta blocks must be staged in pairs");
};

// gamma/beta panel of one head (2*KB out-blocks, 32-wide blocks alternating gamma, beta):
// v[m] = v0[m] * (1 + gamma[m]) + beta[m]   (FiLM, ECA:115-118), stage by stage.
template <int D, bool BF16, int NEXT_BYTES>
DEV void gb_panel(f32x16 (&v)[D / 32], const Frags<BF16, D / 32>& F, Pipe& P, char* ring, unsigned panel,
                  unsigned next, bool active, const float* bias, const float* v0vec, int tid, int lane, int half) {
  using C = typename PairCfg<D, BF16>::GB;
  constexpr int KB = D / 32, MBS = C::MBS;
#pragma unroll
  for (int sp = 0; sp < C::SPP; ++sp) {
    if (sp + 1 < C::SPP) stage_issue<C::STAGE>(P.rs, panel + (sp + 1) * C::STAGE, ring + (P.cur ^ 1) * STAGE_MAX, P.wave, lane);
    else if (next != NO_STAGE) stage_issue<NEXT_BYTES>(P.rs, next, ring + (P.cur ^ 1) * STAGE_MAX, P.wave, lane);
    f32x16 t[MBS];
#pragma unroll
    for (int j = 0; j < MBS; ++j) load_rowvec(t[j], bias, sp * MBS + j, half);
    if (active) gemm_stage<BF16, KB, MBS>(t, F, ring + P.cur * STAGE_MAX, lane);
#pragma unroll
    for (int j = 0; j < MBS / 2; ++j) {
      const int m = sp * (MBS / 2) + j;
      f32x16 v0;
      load_rowvec(v0, v0vec, m, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[m][r] = fmaf(v0[r], 1.0f + t[2 * j][r], t[2 * j + 1][r]);
    }
    stage_wait();
    __syncthreads();
    P.cur ^= 1;
  }
}

// RFF encoding of this lane's pair: E[0..TB) = sin(2 pi t), E[TB..2TB) = cos(2 pi t), t = coeff^T inv
template <int D, bool BF16>
DEV void rff_embed(f32x16 (&E)[D / 32], const float (&inv)[4], const float* cfrag, int lane, int half) {
  constexpr int TB = D / 64;
  const float b0 = half ? inv[1] : inv[0];
  const float b1 = half ? inv[3] : inv[2];
#pragma unroll
  for (int m = 0; m < TB; ++m) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    t = __builtin_amdgcn_mfma_f32_32x32x2f32(cfrag[(m * 2 + 0) * 64 + lane], b0, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x2f32(cfrag[(m * 2 + 1) * 64 + lane], b1, t, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) { E[m][r] = sin_rev<BF16>(t[r]); E[TB + m][r] = cos_rev<BF16>(t[r]); }
  }
}

