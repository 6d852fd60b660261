// enf_layout.h -- byte layout of the packed weight blob and of the workspace (host + device).
//
// The blob holds, for one EquivariantCrossAttentionNeF weight tree:
//   * fp32 row-major matrices/vectors used by the latent prologue (per-latent work, VALU),
//   * fp32 bias / constant vectors used as accumulator initialisers in the per-pair chain,
//   * the per-pair and per-query weight panels in MFMA A-operand fragment order
//     ("panel": 16-row out-tiles x 32-wide in-blocks x 1 KB fragments), bf16 or fp32 (EnfDesc.precision),
//   * the same panels transposed for the backward chain (dX = W dY),
//   * plain fp32 copies of every folded matrix (inputs of the panel packer).
// Folds (exact algebra, fp32; DESIGN.md "Folds"):
//   AF  = rffv.linear_final @ inv_emb_to_v.Dense_0                    (RFF:46 + ECA:17)
//   AGB = diag(LayerNorm_0.scale) @ inv_emb_to_v.Dense_1, per head 32-wide blocks [g b g b ..] (ECA:19-20,115)
//   MU_h = scale * rffq.linear_final @ inv_emb_to_q[:, hD:(h+1)D]      (RFF:46 + ECA:92,134)
//   WB  = blockdiag_h(diag(mixer.LN.scale) @ mixer.Dense_1) @ out_proj @ ffn.Dense_0   (ECA:20,144-150; NEF:66)
//   WF1 = diag(ffn.LN.scale) @ ffn.Dense_1                             (ECA:19-20)
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/enf_hip.h"

#ifdef __HIPCC__
#define ENF_HD __host__ __device__
#else
#define ENF_HD
#endif

struct EnfDims {
  int B, N, Z, H, D, C, O, dx, dp, I, inv, use_window, bf16;
  int Ht;      // the model's true num_heads (== H unless padded with zero heads)
  int Dt;      // the model's true num_hidden (== D unless zero-padded): LayerNorm divisors, logit scale
  int HD;      // H*D
  int KB;      // D/32   in/out blocks of a D-wide activation
  int KBH;     // HD/32
  int OB;      // ceil(O/32) output blocks of the last layer
  // per-call options (EnfDesc): requested pair-kernel variants (ENF_VARIANT_*), relu masks
  int var_fwd, var_bwd, mask_mode, mask_B;
  int mask_b0;   // signal index of b = 0 in the batch the masks were taken for (set by chunked passes; 0 otherwise)
  unsigned* masks;
};

ENF_HD inline int enf_inv_dim(int inv, int dx) {
  switch (inv) {
    case ENF_INV_REL_POS_PERIODIC: return 4;
    case ENF_INV_LATITUDE_PERIODIC: return 4;
    case ENF_INV_POLAR_PERIODIC: return 1;
    case ENF_INV_PONITA: return 2;
    case ENF_INV_PONITA_FULL: return 3;
    case ENF_INV_ABS_POS: return dx;
    case ENF_INV_REL_POS: return dx;
    case ENF_INV_NORM_REL_POS: return 1;
    case ENF_INV_BALL: return 5;
    case ENF_INV_BALL_LAT: return 6;
    default: return -1;
  }
}
ENF_HD inline int enf_inv_pose_dim(int inv, int dx) {
  switch (inv) {
    case ENF_INV_REL_POS_PERIODIC: return 2;
    case ENF_INV_LATITUDE_PERIODIC: return 2;
    case ENF_INV_POLAR_PERIODIC: return 2;
    case ENF_INV_PONITA: return 3;            // (pos_x, pos_y, theta)
    case ENF_INV_PONITA_FULL: return 3;
    case ENF_INV_ABS_POS: return dx;
    case ENF_INV_REL_POS: return dx;
    case ENF_INV_NORM_REL_POS: return dx;
    case ENF_INV_BALL: return 4;              // (alpha, beta, gamma, r)
    case ENF_INV_BALL_LAT: return 4;
    default: return -1;
  }
}

// The kernels carry at most 4 invariants per pair.  ball / ball_lat have 5 / 6, but one / two of them depend on the
// latent only (r_p; th_p and r_p): their contribution coeff[row]^T inv_row to the RFF pre-activation t is a per-latent
// PHASE vector (D/2 values per RFFNet), which the prologue computes and the pair kernels use as the initial value of
// the t accumulator -- zero extra work per pair.  Rows of the coefficient matrix, reference order -> [pair rows | latent rows]:
ENF_HD inline void enf_inv_rows(int inv, int I, int (&pair)[4], int& npair, int (&lat)[2], int& nlat) {
  npair = 0; nlat = 0;
  if (inv == ENF_INV_BALL) { pair[0] = 0; pair[1] = 1; pair[2] = 2; pair[3] = 3; npair = 4; lat[0] = 4; nlat = 1; }
  else if (inv == ENF_INV_BALL_LAT) { pair[0] = 0; pair[1] = 2; pair[2] = 3; pair[3] = 4; npair = 4; lat[0] = 1; lat[1] = 5; nlat = 2; }
  else { for (int i = 0; i < I && i < 4; ++i) pair[npair++] = i; }
}
ENF_HD inline bool enf_inv_has_phase(int inv) { return inv == ENF_INV_BALL || inv == ENF_INV_BALL_LAT; }

inline EnfDims enf_dims(const EnfDesc* d) {
  EnfDims m;
  m.B = d->B; m.N = d->N; m.Z = d->Z; m.H = d->H; m.D = d->D; m.C = d->C; m.O = d->O;
  m.dx = d->dx; m.inv = d->invariant_id; m.use_window = d->use_window;
  m.bf16 = d->precision == ENF_PREC_BF16;
  m.Dt = d->d_true > 0 ? d->d_true : d->D;
  m.Ht = d->h_true > 0 ? d->h_true : d->H;
  m.I = enf_inv_dim(m.inv, m.dx); m.dp = enf_inv_pose_dim(m.inv, m.dx);
  m.HD = m.H * m.D; m.KB = m.D / 32; m.KBH = m.HD / 32; m.OB = (m.O + 31) / 32;
  m.var_fwd = d->pair_fwd_variant; m.var_bwd = d->pair_bwd_variant;
  m.masks = (unsigned*)d->relu_masks;
  m.mask_mode = m.masks ? d->mask_mode : ENF_MASK_OFF;
  m.mask_B = d->mask_signals > 0 ? d->mask_signals : d->B;
  m.mask_b0 = 0;
  return m;
}

// bytes of a packed panel with R output rows (multiple of 16) and K inputs (multiple of 32)
ENF_HD inline size_t enf_panel_bytes(int R, int K, int bf16) { return (size_t)R * K * (bf16 ? 2 : 4); }

struct EnfLayout {
  // ---- prologue, fp32 row-major (in,out)
  size_t stem_w, stem_b, lna_g, lna_b, wk, bk, wv, bv;
  size_t mu;     // H x (D x D): u_h[i] = sum_d mu[h][i][d] * k_h[d]
  size_t cvec;   // H x D:       c_h    = sum_d cvec[h][d] * k_h[d]
  size_t mut;    // H x (D x D): mu transposed per head ([h][d][i]) for the forward prologue's coalesced reads
  size_t wkt, wvt;   // HD x D: a_to_k / a_to_v kernels transposed, for the prologue backward
  // ---- coefficient A-operands of t = coeff^T inv (fp32 16x16x4 MFMA), [D/32 t-tiles][64 lanes]
  size_t acq, acv;
  size_t cphq, cphv;   // 2 x D/2: coefficient rows of the latent-only invariants (ball, ball_lat), zero otherwise
  // ---- accumulator-init vectors, fp32
  size_t bq1, bv1, bf, bgb, bm;         // D, D, D, 2HD (panel order), D
  size_t bB, bF1, bO0, bO2, bO4;        // HD, HD, D, D, 32*OB
  // ---- forward panels (A operand = W^T, out x in)
  size_t aq1, av1, af, agb, am;         // KBxKB each; agb: 2H stages of KBxKB
  size_t atb, atf1, ato0, ato2, ato4;   // KBHxKBH, KBHxKBH, KBxKBH, KBxKB, OBxKB
  // ---- backward panels (A operand = W, in x out): dX = W dY
  size_t gq1, gv1, gf, ggb, gm;         // KBxKB; ggb: H panels of KB x 2KB (out = D, in = that head's [g b ..] 2D)
  size_t gcq, gcv;                      // 1 x D/64: A[c][t] = 2 pi coeff[c][t]  (d inv = 2 pi coeff d t)
  size_t gtb, gtf1, gto0, gto2, gto4;   // KBHxKBH, KBHxKBH, KBHxKB, KBxKB, KBxOB
  // ---- plain fp32 (in,out) copies of folded matrices
  size_t p_af, p_agb, p_wb, p_wf1, p_tmp, p_o4;   // DxD, Dx2HD, HDxHD, HDxHD, HDxHD, Dx(32*OB)
  size_t p_mxw;                         // DxD: diag(mixer LN scale) @ mixer Dense_1
  size_t p_mxb;                         // D: mixer LN bias @ Dense_1 + Dense_1 bias
  // ---- latent-independent parts of the per-latent mixer-input fold (enf_wz.hip)
  size_t p_wbmt;                        // H x (D x D): [h][k][i] = sum_j Wbeta_h[i][j] AM[j][k]
  size_t p_cb;                          // H x D:       [h][k]    = sum_j bbeta_h[j] AM[j][k] + bm[k]
  size_t p_opbg;                        // H x D:       [h][j]    = 1 + bgamma_h[j]
  size_t p_wbm;                         // H x (D x D): p_wbmt transposed ([h][i][k]), for the backward-orientation panels
  size_t awg;                           // H forward panels (D x D): gamma half of inv_emb_to_v.Dense_1, A[j][i] = Wgamma_h[i][j]
  size_t total;
};

inline size_t enf_align(size_t x) { return (x + 255) & ~(size_t)255; }

inline EnfLayout enf_layout(const EnfDims& m) {
  EnfLayout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = enf_align(o + bytes); return r; };
  const size_t f = sizeof(float);
  const int D = m.D, HD = m.HD, H = m.H, C = m.C;
  L.stem_w = take(f * C * D); L.stem_b = take(f * D); L.lna_g = take(f * D); L.lna_b = take(f * D);
  L.wk = take(f * D * HD); L.bk = take(f * HD); L.wv = take(f * D * HD); L.bv = take(f * HD);
  L.mu = take(f * H * D * D); L.cvec = take(f * H * D);
  L.mut = take(f * H * D * D); L.wkt = take(f * HD * D); L.wvt = take(f * HD * D);
  L.acq = take(f * (D / 32) * 64); L.acv = take(f * (D / 32) * 64);   // D/32 t-tiles x 64 lanes
  L.cphq = take(f * D); L.cphv = take(f * D);
  L.bq1 = take(f * D); L.bv1 = take(f * D); L.bf = take(f * D); L.bgb = take(f * 2 * HD); L.bm = take(f * D);
  L.bB = take(f * HD); L.bF1 = take(f * HD); L.bO0 = take(f * D); L.bO2 = take(f * D); L.bO4 = take(f * 32 * m.OB);
  const int bf = m.bf16, OP = 32 * m.OB;
  L.aq1 = take(enf_panel_bytes(D, D, bf)); L.av1 = take(enf_panel_bytes(D, D, bf));
  L.af = take(enf_panel_bytes(D, D, bf)); L.agb = take(enf_panel_bytes(2 * HD, D, bf));
  L.am = take(enf_panel_bytes(D, D, bf));
  L.atb = take(enf_panel_bytes(HD, HD, bf)); L.atf1 = take(enf_panel_bytes(HD, HD, bf));
  L.ato0 = take(enf_panel_bytes(D, HD, bf)); L.ato2 = take(enf_panel_bytes(D, D, bf));
  L.ato4 = take(enf_panel_bytes(OP, D, bf));
  L.gq1 = take(enf_panel_bytes(D, D, bf)); L.gv1 = take(enf_panel_bytes(D, D, bf));
  L.gf = take(enf_panel_bytes(D, D, bf)); L.ggb = take(enf_panel_bytes(D, 2 * HD, bf));
  L.gm = take(enf_panel_bytes(D, D, bf));
  L.gcq = take(enf_panel_bytes(16, D / 2, bf)); L.gcv = take(enf_panel_bytes(16, D / 2, bf));
  L.gtb = take(enf_panel_bytes(HD, HD, bf)); L.gtf1 = take(enf_panel_bytes(HD, HD, bf));
  L.gto0 = take(enf_panel_bytes(HD, D, bf)); L.gto2 = take(enf_panel_bytes(D, D, bf));
  L.gto4 = take(enf_panel_bytes(D, OP, bf));
  L.p_af = take(f * D * D); L.p_agb = take(f * D * 2 * HD); L.p_wb = take(f * HD * HD);
  L.p_wf1 = take(f * HD * HD); L.p_tmp = take(f * HD * HD); L.p_o4 = take(f * D * OP);
  L.p_mxw = take(f * D * D); L.p_mxb = take(f * D);
  L.p_wbmt = take(f * H * D * D); L.p_cb = take(f * H * D); L.p_opbg = take(f * H * D);
  L.p_wbm = take(f * H * D * D); L.awg = take(H * enf_panel_bytes(D, D, bf));
  L.total = o;
  return L;
}

// ---- latent table: one row per (b, z), written by the prologue, read by the pair kernels
// [ u (H*D) | v0 (H*D) | pose (4) | wcoef (1) | pad (3) | c (H) | pad ] fp32, 16-byte aligned fields
// [ u (H*D) | v0 (H*D) | pose (4) | wcoef (1) | pad (3) | c (H <= 4) | pad | ext (16) | phase_q (D/2) | phase_v (D/2) ]
// ext: ball -> the rotation matrix R (9, row-major); in the GRADIENT table: d R (9), then d(latent-only invariants) (2).
// phase_*: the per-latent RFF phase vectors of ball / ball_lat (zeros are never read for the other invariants).
ENF_HD inline int enf_lt_off_ext(int H, int D) { return 2 * H * D + 16; }
ENF_HD inline int enf_lt_off_phq(int H, int D) { return 2 * H * D + 32; }
ENF_HD inline int enf_lt_off_phv(int H, int D) { return 2 * H * D + 32 + D / 2; }
ENF_HD inline int enf_lt_stride(int H, int D) { return ((2 * H * D + 32 + D) + 63) & ~63; }
ENF_HD inline int enf_lt_off_u(int, int) { return 0; }
ENF_HD inline int enf_lt_off_v0(int H, int D) { return H * D; }
ENF_HD inline int enf_lt_off_pose(int H, int D) { return 2 * H * D; }
ENF_HD inline int enf_lt_off_wcoef(int H, int D) { return 2 * H * D + 4; }
ENF_HD inline int enf_lt_off_c(int H, int D) { return 2 * H * D + 8; }

// Forward pair kernel variant.  "z-fold": all 8 waves of a workgroup walk the latents together (one
// latent per step, 128 queries per workgroup), which lets FiLM and the mixer's first Dense collapse
// into ONE per-latent D x D matrix  W_zh = (Wgamma_h diag(v0_zh) + Wbeta_h) AM  (no nonlinearity
// sits between them), built by enf_wz_kernel into `wz` before the pair kernel runs.  It needs
// enough 128-query workgroups to fill the chip; below that the latent-split variant runs.
// EnfDesc.pair_fwd_variant forces the choice per call; ENF_VARIANT_AUTO is the heuristic below, a function of the shape
// alone (only an -DENF_AB_SWITCHES build of the library lets ENF_ZFOLD=0 / 1 in the environment replace it).
int enf_zfold_env(int backward);   // enf_api.hip: -1 (always, in the product library), 0 / 1
// The z-fold kernel's work below 192 query tiles (ENF_VARIANT_ZFOLD_ZSPLIT), "stream-K" over the latents: the flattened
// (signal, 128-query tile, latent) space -- tiles x Z latent steps -- is cut into <= 256 runs of `len` steps, one workgroup each, so a
// single round of workgroups ends together whatever the tile count (144 tiles x 128 latents: 256 runs of 72 -- 0.56 of the unsplit time;
// three equal parts per tile were 432 workgroups = two rounds of a third, 0.67).  A run that crosses a tile boundary is two segments of
// one workgroup; every segment leaves partial softmax sums in its tile's slot (tile t is met by the runs floor(t Z / len) ..
// floor(((t + 1) Z - 1) / len): `parts` = the most any tile has), merged by enf_zsplit_merge_kernel.
struct EnfStreamK { int len, wgs, parts; };        // parts <= 1: not split
inline EnfStreamK enf_streamk(long long tiles, int Z, int len_min) {
  const long long total = tiles * Z;
  long long len = (total + 255) / 256;
  if (len < len_min) len = len_min;
#ifdef ENF_SK_FORCE_LEN      // A/B builds only (scripts/build_variant.sh): e.g. 43 at config 3 = round 3's earlier three equal parts per tile
  len = ENF_SK_FORCE_LEN;
#endif
  EnfStreamK k{(int)len, (int)((total + len - 1) / len), 1};
  if (len >= Z) return k;
  for (long long t = 0; t < tiles; ++t) {
    const int n = (int)((t * Z + Z - 1) / len - (t * Z) / len) + 1;
    if (n > k.parts) k.parts = n;
  }
  return k;
}
// AUTO splits when the 128-query tiles alone would under-fill the chip (< 192) and there are >= 128 latents, with >= 32 latent steps per
// workgroup, when that beats the latent-split kernel by the estimate below.
#ifndef ENF_SK_MIN_Z
#define ENF_SK_MIN_Z 128
#endif
#ifndef ENF_SK_MIN_RUN
#define ENF_SK_MIN_RUN 32
#endif
inline EnfStreamK enf_zfold_streamk(const EnfDims& m) {
  const EnfStreamK none{0, 0, 1};
  const long long tiles = (long long)((m.N + 127) / 128) * m.B;
  if (tiles * m.Z >= 0x7fffffffLL) return none;
  if (m.var_fwd == ENF_VARIANT_ZFOLD_ZSPLIT) return m.Z >= 2 ? enf_streamk(tiles, m.Z, (m.Z + 2) / 3) : none;   // forced: <= 4 parts
  if (m.var_fwd != ENF_VARIANT_AUTO || enf_zfold_env(0) >= 0) return none;
  if (tiles >= 192 || m.Z < ENF_SK_MIN_Z) return none;
  const EnfStreamK k = enf_streamk(tiles, m.Z, ENF_SK_MIN_RUN);
  // against the latent-split kernel: a run costs its latent steps + ~5 (fold kernel, merge), and in the time of one latent step (128
  // pairs on each of 256 CUs at best) the latent-split kernel gets through ~22,000 pairs (measured at D = 128, H = 2: 9.1 us against
  // 0.41 ns per pair).  Config 3 (144 tiles x 128 latents): 77 against 107 -> split; config 4's fit shape (32 tiles x 128 latents: 128
  // runs of 32 would fill half the chip): 37 against 24 -> latent-split, 0.21 against 0.29 ms measured.
  const double latent_split_steps = (double)m.B * m.N * m.Z / 22000.0;
  return k.parts > 1 && k.len + 5 < 0.9 * latent_split_steps ? k : none;
}
// 0 = the latent-split kernel; 1 = the z-fold kernel, one workgroup per 128-query tile walking all latents; s >= 2 = the z-fold kernel over
// equal runs of latent steps, a tile's latents in up to s parts
inline int enf_zfold_split(const EnfDims& m) {
  // one signal's folded matrices sit behind a buffer resource with 32-bit offsets: beyond 2 GB per signal (Z >= 32768
  // at D = 128, H = 2) only the latent-split variant can run
  if ((long long)m.Z * m.H * (long long)m.D * m.D * (m.bf16 ? 2 : 4) >= 0x7fffffffLL) return 0;
  const EnfStreamK k = enf_zfold_streamk(m);
  if (k.parts > 1) return k.parts;
  if (m.var_fwd == ENF_VARIANT_ZFOLD_ZSPLIT) return 1;
  if (m.var_fwd != ENF_VARIANT_AUTO) return m.var_fwd == ENF_VARIANT_ZFOLD ? 1 : 0;
  const int mode = enf_zfold_env(0);
  if (mode >= 0) return mode;
  return (long long)((m.N + 127) / 128) * m.B >= 192 ? 1 : 0;
}
inline bool enf_use_zfold(const EnfDims& m) { return enf_zfold_split(m) > 0; }

ENF_HD inline size_t enf_wzu_bytes(int H, int D) { return (size_t)(D / 32) * 4 * H * 16; }

// Backward counterpart (enf_pair_bwd_kernel<.., ZF = true>): one workgroup per latent, its 8 waves take 8 query
// tiles at a time, so the per-latent matrices W_zh (both orientations) stream through the LDS ring shared by the
// workgroup.  Needs enough latents to fill the chip.  EnfDesc.pair_bwd_variant as for the forward.
inline bool enf_use_zfold_bwd(const EnfDims& m) {
  if (m.var_bwd != ENF_VARIANT_AUTO) return m.var_bwd == ENF_VARIANT_ZFOLD;
  const int mode = enf_zfold_env(1);
  if (mode >= 0) return mode == 1;
  return (long long)m.B * m.Z >= 192;
}

struct EnfWorkspace {
  size_t lt;        // B*Z*lt_stride floats: latent table
  size_t an;        // B*Z*(D + D + 2) floats: stem output, a_norm, LN mean/rstd (prologue backward)
  size_t kv;        // B*Z*2*HD floats: k | v0-as-computed (prologue backward)
  size_t ybar;      // B*N*HD floats (used when the caller passes ybar == NULL)
  size_t lse;       // B*N*H
  size_t dybar;     // B*N*HD  gradient of ybar (backward)
  size_t delta;     // B*N*H   sum_d dybar*ybar (backward)
  size_t tail_act;  // B*N*(2*HD + 2*D + 2) tail pre-activations + LN stats (backward recompute cache)
  size_t dlt;       // B*Z*lt_stride floats: gradient of the latent table (backward)
  size_t wz;        // B*Z*H packed D x D panels: per-latent mixer-input matrices (z-fold forward only)
  size_t wzb;       // B*Z*H*D floats: their bias vectors
  size_t wzt;       // B*Z*H x [forward | backward] packed D x D panels of W_zh (z-fold backward)
  size_t wzu;       // B*Z x (D/32 * 4 * H) x 16 B: the logit vectors u_zh as bf16 A-operand rows (z-fold, bf16 mode)
  size_t ysplit;    // ENF_VARIANT_ZFOLD_ZSPLIT (s splits): s x B*N*HD partial weighted sums | s x B*N*H x 3 softmax statistics (m, l, c)
  size_t total;
};

inline EnfWorkspace enf_workspace(const EnfDims& m) {
  EnfWorkspace W;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = enf_align(o + bytes); return r; };
  const size_t f = sizeof(float);
  const size_t BZ = (size_t)m.B * m.Z, BN = (size_t)m.B * m.N;
  W.lt = take(f * BZ * enf_lt_stride(m.H, m.D));
  W.an = take(f * BZ * (2 * m.D + 2));
  W.kv = take(f * BZ * 2 * m.HD);
  W.ybar = take(f * BN * m.HD);
  W.lse = take(f * BN * m.H);
  W.dybar = take(f * BN * m.HD);
  W.delta = take(f * BN * m.H);
  W.tail_act = take(f * BN * (2 * m.HD + 2 * m.D + 2));
  W.dlt = take(f * BZ * enf_lt_stride(m.H, m.D));
  const bool zf = enf_use_zfold(m), zb = enf_use_zfold_bwd(m);
  W.wz = take(zf || zb ? BZ * m.H * enf_panel_bytes(m.D, m.D, m.bf16) : 0);
  W.wzb = take(zf || zb ? f * BZ * m.HD : 0);
  W.wzt = take(zb ? BZ * m.H * 2 * enf_panel_bytes(m.D, m.D, m.bf16) : 0);
  W.wzu = take(zf ? BZ * enf_wzu_bytes(m.H, m.D) : 0);
  W.ysplit = take(enf_zfold_split(m) > 1 ? f * enf_zfold_split(m) * (BN * m.HD + BN * m.H * 3) : 0);
  W.total = o;
  return W;
}
