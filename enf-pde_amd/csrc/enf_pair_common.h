// enf_pair_common.h -- pieces shared by the forward (K2) and backward (K3) pair kernels.
#pragma once
#include "enf_device.h"

template <int D, bool BF16> struct PairCfg {
  static constexpr int KB = D / 32;             // 32-feature blocks of a D-wide activation
  static constexpr int NT = D / 16;             // 16-feature tiles
  using DD = PanelCfg<KB, NT, BF16>;            // a D x D panel
  using GB = PanelCfg<KB, 2 * NT, BF16>;        // one head's gamma/beta panel (2D outputs)
  static_assert(GB::MTS % 4 == 0, "gamma/beta tiles must be staged in [g g b b] groups");
};

// gamma/beta panel of one head: 2D outputs ordered per 32-feature block m as
// [gamma tiles 2m, 2m+1 | beta tiles 2m, 2m+1] (reorder_gb_kernel).  FiLM (ECA:115-118):
//   v = v0 * (1 + gamma) + beta, stage by stage; OPG (optional) returns 1 + gamma.
template <int D, bool BF16, int NEXT_BYTES, bool KEEP, int NW = NWAVES>
DEV void gb_panel(f32x4 (&v)[D / 16], f32x4 (&opg)[KEEP ? D / 16 : 1], const Frags<BF16, D / 32>& F, Pipe& P, char* ring,
                  unsigned panel, unsigned next, bool active, const float* bias, const float* v0vec, int lane, int quad) {
  using C = typename PairCfg<D, BF16>::GB;
  constexpr int KB = D / 32, MTS = C::MTS;
#pragma unroll
  for (int sp = 0; sp < C::SPP; ++sp) {
    stage_open(P);
    if (sp + 1 < C::SPP) stage_issue_p<C::STAGE, NW>(P, panel + (sp + 1) * C::STAGE, ring + (P.cur ^ 1) * STAGE_MAX, lane);
    else if (next != NO_STAGE) stage_issue_p<NEXT_BYTES, NW>(P, next, ring + (P.cur ^ 1) * STAGE_MAX, lane);
    f32x4 t[MTS];
#pragma unroll
    for (int j = 0; j < MTS; ++j) t[j] = rowvec(bias, sp * MTS + j, quad);
    if (active) gemm_stage<BF16, KB, MTS>(t, F, ring + P.cur * STAGE_MAX, lane);
#pragma unroll
    for (int g = 0; g < MTS / 4; ++g) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int tile = 2 * (sp * (MTS / 4) + g) + e;
        const f32x4 v0 = rowvec(v0vec, tile, quad);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float og = 1.0f + t[4 * g + e][i];
          if constexpr (KEEP) opg[tile][i] = og;
          v[tile][i] = fmaf(v0[i], og, t[4 * g + 2 + e][i]);
        }
      }
    }
    stage_close(P);
  }
}

// RFF encoding of this lane's pair: E[0..D/32) = sin(2 pi t), E[D/32..D/16) = cos(2 pi t),
// t = coeff^T inv by one fp32 MFMA per 16 t-values (K = 4 = the invariant components)   RFF:86-93
// `phase` (fp32 in LDS, D/2 values, or nullptr): the per-latent part of t for invariants with latent-only components
template <int D, bool BF16>
DEV void rff_embed(f32x4 (&E)[D / 16], const float (&inv)[4], const float* cfrag, int lane, int quad, const float* phase = nullptr) {
  constexpr int TT = D / 32;
  const float bq = quad == 0 ? inv[0] : quad == 1 ? inv[1] : quad == 2 ? inv[2] : inv[3];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (phase) t = *reinterpret_cast<const f32x4*>(phase + 16 * tt + 4 * quad);
    t = __builtin_amdgcn_mfma_f32_16x16x4f32(cfrag[tt * 64 + lane], bq, t, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { E[tt][i] = sin_rev<BF16>(t[i]); E[TT + tt][i] = cos_rev<BF16>(t[i]); }
  }
}
