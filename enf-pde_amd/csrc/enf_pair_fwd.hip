// enf_pair_fwd.hip -- K2: fused per-pair chain + cross-attention over the latent set.
//
// One workgroup = 8 waves (2 per SIMD).  A wave owns 16 queries (columns) of one signal and a
// share of the Z latents: the 8 waves form QG query groups x ZS latent splits (ZS = 8 for Z >= 8,
// so a workgroup covers 16 queries; fewer splits / more query groups for tiny Z).  For each
// (16 queries) x (one latent) the wave runs, entirely in registers and in the transposed
// "acc layout" of enf_device.h:
//   invariant (INV/*)                -> inv (I<=4)                              ECA:86
//   t = coeff^T inv (fp32 MFMA), [sin,cos]                                      RFF:86-93
//   h1 = relu(W1q^T e + b)           (query RFFNet layer)                       RFF:63-64
//   logit_h = h1.u_h + c_h + window  (RFF linear_final, inv_emb_to_q and the q.k dot folded
//                                     into the per-latent vector u_h)            RFF:46, ECA:92,134,139
//   g1 = relu(W1v^T e_v + b); f = gelu(AF^T g1 + b); n = LayerNorm(f)           RFF:63-64,46; ECA:17-19
//   [gamma_h; beta_h] = AGB^T n + b ; v_h = v0_h (1+gamma_h) + beta_h           ECA:20,115-121
//   g = gelu(AM^T v_h + b); (mu, rstd) = LN stats of g                          ECA:122 -> ECA:17-19
//   softmax over z of logit_h; ybar_h += softmax * (g - mu) * rstd              ECA:141-144
// The mixer's LayerNorm affine and Dense_1, attn.out_proj and the block FFN's Dense_0 are
// linear in the softmax-weighted sum and are applied once per query by the tail kernel.
// Weight panels stream L2 -> LDS by LDS-DMA through a 2-slot ring shared by the 8 waves.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_launch.h"
#include "enf_device.h"
#include "enf_pair_common.h"

struct PairFwdArgs {
  const float* x; long long x_bstride;
  const float* lt; const char* blob; EnfLayout L;
  float* ybar; float* lse;
  const char* wz; const float* wzb; const char* wzu;   // z-fold only: per-latent mixer-input panels / biases (enf_wz.hip)
  float inv_d;                            // 1 / (true num_hidden)
  int xcd_remap;                          // z-fold: 1 when B % 8 == 0 (see the kernel)
  unsigned* masks; int mask_mode, mask_B; // relu masks (ENF_MASK_*): buffer, 0 off / 1 write / 2 read, signals per mask set
  int B, N, Z, dx, inv, use_window, qg;   // qg: query groups per workgroup (1,2,4,8); ZS = 8/qg
  int zsplit;                             // z-fold, ENF_VARIANT_ZFOLD_ZSPLIT: the most parts a query tile's latents are cut into (1: no split)
  int sk_len;                             // zsplit > 1: latent steps per workgroup -- workgroup c walks steps [c sk_len, (c + 1) sk_len) of the
                                          // flattened (signal, query tile, latent) space (enf_layout.h: enf_zfold_streamk)
  int ybar_half;                          // ENF_STAGE_YBAR_HALF: `ybar` is written as bf16 rows (H D x 2 bytes) for the tail kernel of the same call
  float* ysplit;                          // zsplit > 1: [zsplit][B N HD] partial sums | [zsplit][B N H][3] (m, l, c), merged by enf_zsplit_merge_kernel
};

// Debug build only (-DENF_STAMPS): s_memtime stamps of the first iterations of workgroup 0, one row per
// wave, read back with enf_debug_read_stamps(); the buffer is read by no kernel code (MI355X_MICROARCH.md).
#ifdef ENF_STAMPS
__device__ unsigned long long enf_stamps[8 * 4 * 24];
#define STAMP(k)                                                                                              \
  do {                                                                                                        \
    if (blockIdx.x == 7 && blockIdx.y == 0 && lane == 0 && it < 4)                                            \
      enf_stamps[(wave * 4 + it) * 24 + (k)] = __builtin_amdgcn_s_memtime();                                  \
  } while (0)
extern "C" int enf_debug_read_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(enf_stamps), sizeof(enf_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define STAMP(k) do {} while (0)
#endif

#ifndef ENF_K2_BIAS_IN_STAGE
#define ENF_K2_BIAS_IN_STAGE 0   // measured: 1.19 vs 1.135 ms with the bias loads inside the asm stage (K2 has registers to spare)
#endif
#define K2_INIT (ENF_K2_BIAS_IN_STAGE ? INIT_BIAS : INIT_ACC)
// accumulators start from the bias row vector: loaded here (INIT_ACC) or inside the asm stage (INIT_BIAS)
#define K2_BIAS(ACC, PTR)                                                              \
  do {                                                                                 \
    if (K2_INIT == INIT_ACC) {                                                         \
      _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) ACC[t_] = rowvec(PTR, t_, quad); \
    }                                                                                  \
  } while (0)
// relu-mask modes live in their own instantiation (MASKS): as wave-uniform runtime branches in the one kernel they cost
// the default path 1.3 % (measured)
#define K2_MASK_MODE (MASKS ? A.mask_mode : 0)
#ifndef ENF_ZFOLD_WAVES
#define ENF_ZFOLD_WAVES 8
#endif
#ifndef ENF_K2_LA             // z-fold bf16: look-ahead staging (enf_device.h: panel_gemm<.., LA>).  OFF: measured 3 % SLOWER on this
#define ENF_K2_LA 0           // kernel (decode shape 1.275 vs 1.238 ms same-box, gpurun_out/r02/ab_la3.log): every stage here is followed
#endif                        // by a vector epilogue longer than the DMA, which the 2-slot order already hides; in K3 it pays (-7 %)
#ifndef ENF_ANTIPHASE
#define ENF_ANTIPHASE false
#endif
#ifndef ENF_K2_A3             // z-fold bf16: antiphase staging over a three-slot ring (enf_device.h: panel_gemm_a3).  OFF: correct (189 forward /
#define ENF_K2_A3 0           // golden tests) but 2 % SLOWER at the decode shape (1.153 vs 1.129 ms same-box, profiles/r03_ab_k2_a3.log): an MFMA
#endif                        // holds the SIMD's vector issue for half its duration, so the mate's epilogue does not run "under" it for free
#ifndef ENF_K2_LN_ASM         // the LayerNorm apply as scalar asm fmas (ln_apply, enf_device.h: the form K3 needed).  K2's fmaf loop packed
#define ENF_K2_LN_ASM 1       // differently and never deviated (scripts/k3_race/fwd_probe.py); the asm form costs nothing (same-box A/B: decode
#endif                        // 1.238-1.241 vs 1.240-1.250 ms), so K2 avoids the instruction class too

// waves per workgroup: the z-fold variant runs TWO independent 4-wave workgroups per CU (one wave of each
// per SIMD) so that the SIMD-mates never meet at a barrier: while one computes its MFMA stage the other
// keeps the vector ALU busy (8-wave lockstep measured 2 V + M per stage, M = the younger wave's MFMAs).
#ifndef ENF_K2_INV_SPECIALISED
#define ENF_K2_INV_SPECIALISED 1
#endif
template <bool ZFOLD> struct PairWaves { static constexpr int NW = ZFOLD && ENF_ZFOLD_WAVES == 4 ? 4 : NWAVES; };

template <int D, int H, bool BF16, int NW, int NSLOT = 2> struct PairSmem {
  static constexpr int RING = 0;                                   // 2 slots (3: A3 staging)
  static constexpr int CONSTS = RING + NSLOT * STAGE_MAX;              // bq1 bv1 bf bm (D each) | bgb (2HD) | acq acv (2D each)
  static constexpr int N_CONST = 4 * D + 2 * H * D + 4 * D;
  static constexpr int ZVEC = CONSTS + 4 * N_CONST;                // NWAVES x 2*H*D floats
  static constexpr int XCH = ZVEC + 4 * NW * 2 * H * D;            // NW x H x 3 x 16 floats
  static constexpr int TOTAL = XCH + 4 * NW * H * 3 * 16;
  static constexpr int YBYTES = H * (D / 16) * 4 * 64 * 4;         // one wave's Y in [reg][lane] order
  static_assert(4 * YBYTES <= 2 * STAGE_MAX, "combine buffer must fit in the ring");
};

// ZFOLD: qg = 8 (every wave walks all latents, the 8 waves in step), and per head the gamma/beta GEMM, FiLM
// and the mixer's first Dense are ONE D x D GEMM with the per-latent matrix W_zh of enf_wz.hip:
//   a5_h = W_zh^T n + c_zh      (5 D x D GEMMs per pair instead of 9 D x D equivalents)
// INV >= 0: the invariant as a compile-time constant (the shipped configs' instantiations, launch code below): the per-latent step then
// carries no switch over the invariant (as in K3, enf_pair_bwd.hip)
template <int D, int H, bool BF16, bool ZFOLD, bool MASKS, int INV = -1>
__global__ __launch_bounds__(64 * PairWaves<ZFOLD>::NW, 2) void enf_pair_fwd_kernel(PairFwdArgs A) {
  const int inv_id = INV >= 0 ? INV : A.inv;
  const int dx_ = INV >= 0 ? 2 : A.dx;
  using Cfg = PairCfg<D, BF16>;
  constexpr int NW = PairWaves<ZFOLD>::NW, NTH = 64 * NW;
  constexpr bool A3 = ZFOLD && BF16 && ENF_K2_A3 != 0 && NW == 8 && PairCfg<D, BF16>::DD::SPP == 1;
  using SM = PairSmem<D, H, BF16, NW, A3 ? 3 : 2>;
  constexpr int KB = Cfg::KB, NT = Cfg::NT;
  constexpr int ST_DD = Cfg::DD::STAGE, ST_GB = Cfg::GB::STAGE, PANEL_GB = Cfg::GB::BYTES, PANEL_DD = Cfg::DD::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem + SM::RING;
  float* cst = reinterpret_cast<float*>(smem + SM::CONSTS);
  float* c_bq1 = cst, *c_bv1 = cst + D, *c_bf = cst + 2 * D, *c_bm = cst + 3 * D, *c_bgb = cst + 4 * D;
  float* c_acq = c_bgb + 2 * H * D, *c_acv = c_acq + 2 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, quad = lane >> 4;
  float* zv = reinterpret_cast<float*>(smem + SM::ZVEC) + wave * 2 * H * D;
  float* xch = reinterpret_cast<float*>(smem + SM::XCH);
  const int QG = A.qg, ZS = NW / QG;
  const int qgi = wave % QG, zs = wave / QG;
  // XCD-aware placement (z-fold, B % 8 == 0): workgroup ids go round-robin over the 8 XCDs, and every workgroup of
  // a signal streams that signal's per-latent panels, so all of a signal's workgroups are given ids of ONE residue
  // mod 8: its 64 KB x Z of panels then enter one XCD's L2 instead of eight (HBM/fabric fetches 540 -> ~70 MB).
  int bx = blockIdx.x, by = blockIdx.y;
  if (ZFOLD && A.xcd_remap) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id & 7, slot = id >> 3;
    by = xcd + 8 * (slot / (int)gridDim.x);
    bx = slot % (int)gridDim.x;
  }
  // ENF_VARIANT_ZFOLD_ZSPLIT ("stream-K" over the latents): this workgroup owns the latent steps [f0, f1) of the flattened
  // (signal, query tile, latent) space -- the same count for every workgroup, so one round of <= 256 workgroups ends together -- and
  // walks them as one or more SEGMENTS, each a run of latents of one query tile that leaves partial sums in that tile's slot `part`
  const bool split = ZFOLD && A.zsplit > 1;
  const int tiles_n = (A.N + 16 * QG - 1) / (16 * QG);
  int f0 = split ? (int)blockIdx.x * A.sk_len : 0;
  const int f1 = split ? min(f0 + A.sk_len, tiles_n * A.B * A.Z) : 0;
  const char* blob = A.blob;
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };

  // ---- constants -> LDS
  for (int i = tid; i < D; i += NTH) { c_bq1[i] = G(A.L.bq1)[i]; c_bv1[i] = G(A.L.bv1)[i]; c_bf[i] = G(A.L.bf)[i]; c_bm[i] = G(A.L.bm)[i]; }
  for (int i = tid; i < 2 * H * D; i += NTH) c_bgb[i] = G(A.L.bgb)[i];
  for (int i = tid; i < 2 * D; i += NTH) { c_acq[i] = G(A.L.acq)[i]; c_acv[i] = G(A.L.acv)[i]; }

  const unsigned pQ1 = (unsigned)A.L.aq1, pV1 = (unsigned)A.L.av1, pF = (unsigned)A.L.af, pGB = (unsigned)A.L.agb, pM = (unsigned)A.L.am;
  Pipe P;
  P.rs = make_blob_rsrc(blob, (unsigned)A.L.total);
  float sm_m[H], sm_l[H], sm_c[H];     // softmax state against a per-column reference logit (the first one seen); fp32 accumulators
  f32x4 Y[H][NT];
  int b, n0;
 for (;;) {                            // one pass per segment (exactly one without the split)
  int z_lo = 0, seg_iters = 0, part = 0;
  if (split) {
    const int tf = f0 / A.Z;
    z_lo = f0 - tf * A.Z;
    seg_iters = min(A.Z - z_lo, f1 - f0);
    part = (int)blockIdx.x - (tf * A.Z) / A.sk_len;
    by = tf / tiles_n;
    bx = tf - by * tiles_n;
    f0 += seg_iters;
  }
  b = by;
  n0 = (bx * QG + qgi) * 16;
  const int n = min(n0 + col, A.N - 1);
  const QueryPt q = load_query(A.x + (size_t)b * A.x_bstride + (size_t)n * dx_, dx_, inv_id);
  if constexpr (ZFOLD) P.rs2 = make_blob_rsrc(A.wz + (size_t)b * A.Z * H * PANEL_DD, (unsigned)(A.Z * H * PANEL_DD));
  else P.rs2 = P.rs;
  if constexpr (A3) first_stage_a3<ST_DD, ST_DD, NW>(P, ring, pQ1, pV1, wave, lane);
  else first_stage<ST_DD, NW, ENF_ANTIPHASE>(P, ring, pQ1, wave, lane);
  // look-ahead staging: all five panels of a z-fold bf16 iteration are single 32 KB (8 KB) stages, so the stage after next is
  // issued behind each stage's closing barrier and streams under the vector epilogue that follows every stage of this kernel;
  // call sites pass `LA ? <stage after next> : <next stage>`
  constexpr bool LA = A3 || (ZFOLD && BF16 && ENF_K2_LA != 0 && !ENF_ANTIPHASE && Cfg::DD::SPP == 1);   // (call sites name the stage AFTER next)
  if constexpr (LA && !A3) stage_issue_p<ST_DD, NW>(P, pV1, ring + STAGE_MAX, lane);
  // the z-fold stages of this kernel: 2-slot (optionally look-ahead) staging, or the antiphase 3-slot form
  auto zgemm = [&](f32x4 (&acc_)[NT], const Frags<BF16, KB>& F_, unsigned panel_, unsigned next_, bool active_, const float* bias_) {
    if constexpr (A3) panel_gemm_a3<KB, NT, BF16, ST_DD, NW, K2_INIT>(acc_, F_, P, ring, next_, active_, lane, bias_);
    else panel_gemm<KB, NT, BF16, ST_DD, NW, K2_INIT, LA>(acc_, F_, P, ring, panel_, next_, active_, lane, bias_);
  };

#pragma unroll
  for (int h = 0; h < H; ++h) {
    sm_m[h] = -INFINITY; sm_l[h] = 0.f; sm_c[h] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) Y[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int ltstride = enf_lt_stride(H, D);
  const int iters = split ? seg_iters : ZFOLD ? A.Z : (A.Z + ZS - 1) / ZS;
  for (int it = 0; it < iters; ++it) {
    const int z = ZFOLD ? z_lo + it : it * ZS + zs;
    const bool active = z < A.Z;
    const float* ltrow = A.lt + ((size_t)b * A.Z + (active ? z : A.Z - 1)) * ltstride;
    STAMP(0);
    // per-latent vectors u | v0 -> wave-private LDS
    if constexpr (ZFOLD) {      // u (bf16: as A-operand rows) and c_zh from the fold kernel
      const float* czrow = A.wzb + ((size_t)b * A.Z + z) * (H * D);
#pragma unroll
      for (int i = lane * 4; i < H * D; i += 256) {
        if constexpr (!BF16) *reinterpret_cast<f32x4*>(zv + i) = *reinterpret_cast<const f32x4*>(ltrow + i);
        *reinterpret_cast<f32x4*>(zv + H * D + i) = *reinterpret_cast<const f32x4*>(czrow + i);
      }
      if constexpr (BF16) {
        if (lane < KB * 4 * H)
          *reinterpret_cast<f32x4*>(zv + lane * 4) =
              *reinterpret_cast<const f32x4*>(A.wzu + ((size_t)b * A.Z + z) * enf_wzu_bytes(H, D) + lane * 16);
      }
    } else {
#pragma unroll
      for (int i = lane * 4; i < 2 * H * D; i += 256)
        *reinterpret_cast<f32x4*>(zv + i) = *reinterpret_cast<const f32x4*>(ltrow + i);
    }
    const f32x4 pz = *reinterpret_cast<const f32x4*>(ltrow + enf_lt_off_pose(H, D));
    const float wcoef = ltrow[enf_lt_off_wcoef(H, D)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float inv[4], win;
    const bool has_ph = INV < 0 && D == 64 && enf_inv_has_phase(inv_id);   // ball / ball_lat (64-wide only): rotation matrix and RFF phases of the latent
    pair_invariant<BF16>(inv_id, dx_, q, pz, wcoef, A.use_window, inv, win, ltrow + enf_lt_off_ext(H, D));

    float logit[H];
    Frags<BF16, KB> F;
    {  // ---------------- query branch
      f32x4 acc[NT];
      rff_embed<D, BF16>(acc, inv, c_acq, lane, quad, has_ph ? ltrow + enf_lt_off_phq(H, D) : nullptr);
      make_frags<BF16, KB>(F, acc);
      K2_BIAS(acc, c_bq1);
      STAMP(1);
      if constexpr (ZFOLD) zgemm(acc, F, pQ1, LA ? pF : pV1, active, c_bq1);
      else panel_gemm<KB, NT, BF16, ST_DD, NW, K2_INIT, LA>(acc, F, P, ring, pQ1, LA ? pF : pV1, active, lane, c_bq1);
      STAMP(2);
      const bool mread = K2_MASK_MODE == 2;                       // wave-uniform
      if (K2_MASK_MODE) {
        const size_t mrow = relu_mask_index(b % A.mask_B, A.Z, active ? z : A.Z - 1, (A.N + 15) / 16, n0 / 16, 0, lane);
        if (K2_MASK_MODE == 1 && active && n0 < A.N) A.masks[mrow] = relu_mask_of<NT>(acc);
        if (mread) relu_apply_mask<NT>(acc, n0 < A.N ? A.masks[mrow] : 0u);
      }
      if constexpr (ZFOLD && BF16) {
        // logits on the matrix pipe: rows 0..H-1 of the A operand are u_zh (bf16, packed by enf_wz_kernel),
        // B = relu(a1) fragments; lane (col, quad 0) register h then holds h1 . u_h of query `col`
        static_assert(H <= 4, "logit rows live in one quad");
        make_frags<BF16, KB>(F, acc);
        if (!mread) relu_frags<BF16, KB>(F);
        f32x4 lg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
          bf16x8 ua = __builtin_bit_cast(bf16x8, f32x4{0.f, 0.f, 0.f, 0.f});
          if (col < H) ua = *reinterpret_cast<const bf16x8*>(zv + ((blk * 4 + quad) * H + col) * 4);
          lg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua, F.f[blk], lg, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) logit[h] = __shfl(lg[h], col, 64) + ltrow[enf_lt_off_c(H, D) + h] + win;
      } else
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float s = 0.f;           // (as a packed-pair sum -- tiles_dot, enf_device.h -- this loop measured 0.6 % SLOWER on the fit-shape forward)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 u = rowvec(zv + h * D, t, quad);
#pragma unroll
          for (int i = 0; i < 4; ++i) s = fmaf(mread ? acc[t][i] : relu_f(acc[t][i]), u[i], s);
        }
        logit[h] = xquad_sum(s) + ltrow[enf_lt_off_c(H, D) + h] + win;
      }
    }
    STAMP(3);
    {  // ---------------- value branch: RFFNet layer, folded (linear_final . Dense_0), gelu, LN
      f32x4 acc[NT];
      rff_embed<D, BF16>(acc, inv, c_acv, lane, quad, has_ph ? ltrow + enf_lt_off_phv(H, D) : nullptr);
      make_frags<BF16, KB>(F, acc);
      K2_BIAS(acc, c_bv1);
      STAMP(4);
      if constexpr (ZFOLD) zgemm(acc, F, pV1, LA ? STAGE_RS2 | (unsigned)(z * H * PANEL_DD) : pF, active, c_bv1);
      else panel_gemm<KB, NT, BF16, ST_DD, NW, K2_INIT, LA>(acc, F, P, ring, pV1, pF, active, lane, c_bv1);
      STAMP(5);
      const bool mread = K2_MASK_MODE == 2;
      if (K2_MASK_MODE) {
        const size_t mrow = relu_mask_index(b % A.mask_B, A.Z, active ? z : A.Z - 1, (A.N + 15) / 16, n0 / 16, 1, lane);
        if (K2_MASK_MODE == 1 && active && n0 < A.N) A.masks[mrow] = relu_mask_of<NT>(acc);
        if (mread) relu_apply_mask<NT>(acc, n0 < A.N ? A.masks[mrow] : 0u);
      }
      make_frags<BF16, KB>(F, acc);
      if (!mread) relu_frags<BF16, KB>(F);
      K2_BIAS(acc, c_bf);
      STAMP(6);
      if constexpr (ZFOLD) {
        const unsigned wz0 = STAGE_RS2 | (unsigned)(z * H * PANEL_DD);
        const unsigned after = H > 1 ? wz0 + PANEL_DD : (it + 1 < iters ? pQ1 : NO_STAGE);
        zgemm(acc, F, pF, LA ? after : wz0, active, c_bf);
      }
      else panel_gemm<KB, NT, BF16, ST_GB, NW, K2_INIT>(acc, F, P, ring, pF, pGB, active, lane, c_bf);
      STAMP(7);
      gelu_tiles<NT, BF16>(acc);
      float mu, rstd;
      ln_stats<NT>(acc, mu, rstd, A.inv_d);
#if ENF_K2_LN_ASM
      ln_apply<NT>(acc, mu, rstd);
#else
      const float nmr = -mu * rstd;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = fmaf(acc[t][i], rstd, nmr);
#endif
      make_frags<BF16, KB>(F, acc);   // F = normalised f, shared by all heads' gamma/beta panels
      STAMP(8);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      f32x4 v[NT];
      if constexpr (ZFOLD) {
        const unsigned wzh = STAGE_RS2 | (unsigned)((z * H + h) * PANEL_DD);
        K2_BIAS(v, zv + H * D + h * D);
        STAMP(10 + 4 * h);
        const bool more = it + 1 < iters;
        const unsigned nx1 = h + 1 < H ? wzh + PANEL_DD : (more ? pQ1 : NO_STAGE);
        const unsigned nx2 = h + 2 < H ? wzh + 2 * PANEL_DD : (h + 2 == H ? (more ? pQ1 : NO_STAGE) : (more ? pV1 : NO_STAGE));
        zgemm(v, F, wzh, LA ? nx2 : nx1, active, zv + H * D + h * D);
      } else {
        f32x4 dummy[1];
        gb_panel<D, BF16, ST_DD, false, NW>(v, dummy, F, P, ring, pGB + h * PANEL_GB, pM, active, c_bgb + 2 * h * D,
                                        zv + H * D + h * D, lane, quad);
        STAMP(9 + 4 * h);
        Frags<BF16, KB> FV;
        make_frags<BF16, KB>(FV, v);
        K2_BIAS(v, c_bm);
        STAMP(10 + 4 * h);
        if (h + 1 < H) panel_gemm<KB, NT, BF16, ST_GB, NW, K2_INIT>(v, FV, P, ring, pM, pGB + (h + 1) * PANEL_GB, active, lane, c_bm);
        else panel_gemm<KB, NT, BF16, ST_DD, NW, K2_INIT>(v, FV, P, ring, pM, it + 1 < iters ? pQ1 : NO_STAGE, active, lane, c_bm);
      }
      STAMP(11 + 4 * h);
      gelu_tiles<NT, BF16>(v);
      float mu, rstd;
      ln_stats<NT>(v, mu, rstd, A.inv_d);
      if (active) {
        // softmax over this wave's latents (ECA:141-144).  The rare "logit far above the
        // reference" case rescales the accumulators (wave-uniform branch).
        if (it == 0) sm_m[h] = logit[h];
        const bool far = logit[h] - sm_m[h] > 40.0f;
        if (__any(far)) {
          const float alpha = far ? __expf(sm_m[h] - logit[h]) : 1.0f;
          sm_l[h] *= alpha; sm_c[h] *= alpha;
          if (far) sm_m[h] = logit[h];
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) Y[h][t][i] *= alpha;
        }
        const float pe = __expf(logit[h] - sm_m[h]);
        const float w = pe * rstd;
        sm_l[h] += pe;
        sm_c[h] = fmaf(w, mu, sm_c[h]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) Y[h][t][i] = fmaf(w, v[t][i], Y[h][t][i]);
      }
      STAMP(12 + 4 * h);
    }
  }

  if constexpr (!A3) pipe_finish(P);       // (A3: both halves have met every barrier)
  else __syncthreads();                    // the ring is reused below
  if constexpr (ZFOLD) {
    if (split) {               // partial weighted sums against this segment's own reference logit + (m, l, c): merged afterwards
      if (n0 + col < A.N) {
        const size_t BN = (size_t)A.B * A.N, row = (size_t)b * A.N + n0 + col;
        float* yo = A.ysplit + ((size_t)part * BN + row) * (H * D);
        float* so = A.ysplit + (size_t)A.zsplit * BN * (H * D) + ((size_t)part * BN + row) * (H * 3);
#pragma unroll
        for (int h = 0; h < H; ++h) {
#pragma unroll
          for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(yo + h * D + 16 * t + 4 * quad) = Y[h][t];
          if (quad == 0) { so[h * 3] = sm_m[h]; so[h * 3 + 1] = sm_l[h]; so[h * 3 + 2] = sm_c[h]; }
        }
      }
      if (f0 >= f1) return;
      continue;                // (the closing barrier of the last stage is behind every wave: the ring and zv are free again)
    }
  }
  break;
 }
  // ---- combine the ZS latent splits of each query group (all staging is finished: ring is free)
  if (quad == 0) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      xch[((wave * H + h) * 3 + 0) * 16 + col] = sm_m[h];
      xch[((wave * H + h) * 3 + 1) * 16 + col] = sm_l[h];
      xch[((wave * H + h) * 3 + 2) * 16 + col] = sm_c[h];
    }
  }
  __syncthreads();
  float Ltot[H], Ctot[H], mstar[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float ms = -INFINITY;
    for (int s = 0; s < ZS; ++s) ms = fmaxf(ms, xch[(((s * QG + qgi) * H + h) * 3 + 0) * 16 + col]);
    float L = 0.f, C = 0.f;
    for (int s = 0; s < ZS; ++s) {
      const int w = s * QG + qgi;
      const float aw = __expf(xch[((w * H + h) * 3 + 0) * 16 + col] - ms);
      L = fmaf(aw, xch[((w * H + h) * 3 + 1) * 16 + col], L);
      C = fmaf(aw, xch[((w * H + h) * 3 + 2) * 16 + col], C);
    }
    mstar[h] = ms; Ltot[h] = L; Ctot[h] = C;
    const float sc = __expf(sm_m[h] - ms) / L;     // exp(-inf) = 0 for a split that saw no latent
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) Y[h][t][i] *= sc;
  }
  // deterministic tree over the split index: splits [s, 2s) hand their partial sums to [0, s)
  float* cb = reinterpret_cast<float*>(ring);
  for (int s = ZS >> 1; s >= 1; s >>= 1) {
    if (zs >= s && zs < 2 * s) {
      float* dst = cb + (size_t)((zs - s) * QG + qgi) * (SM::YBYTES / 4);
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[((h * NT + t) * 4 + i) * 64 + lane] = Y[h][t][i];
    }
    __syncthreads();
    if (zs < s) {
      const float* src = cb + (size_t)(zs * QG + qgi) * (SM::YBYTES / 4);
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) Y[h][t][i] += src[((h * NT + t) * 4 + i) * 64 + lane];
    }
    __syncthreads();
  }
  if (zs == 0 && n0 + col < A.N) {
    float* yo = A.ybar + ((size_t)b * A.N + n0 + col) * (H * D);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float cs = Ctot[h] / Ltot[h];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 o = Y[h][t];
        o[0] -= cs; o[1] -= cs; o[2] -= cs; o[3] -= cs;
        if (!MASKS && BF16 && A.ybar_half) {      // (wave-uniform; masked passes are training passes) round to nearest even, as the tail's make_frags would
          unsigned short* yh = reinterpret_cast<unsigned short*>(A.ybar) + ((size_t)b * A.N + n0 + col) * (H * D);
          *reinterpret_cast<uint2*>(yh + h * D + 16 * t + 4 * quad) = uint2{bf16_pack2(o[0], o[1]), bf16_pack2(o[2], o[3])};
        } else {
          *reinterpret_cast<f32x4*>(yo + h * D + 16 * t + 4 * quad) = o;
        }
      }
      if (quad == 0) A.lse[((size_t)b * A.N + n0 + col) * H + h] = mstar[h] + __logf(Ltot[h]);
    }
  }
}

// ENF_VARIANT_ZFOLD_ZSPLIT: ybar[row][h][:] = sum_s e^{m_s - m*} Y_s / L - C / L,  lse = m* + log L  with  L = sum_s e^{m_s - m*} l_s (C alike):
// exactly the in-kernel combine of the latent-split variant, across workgroups.  One thread per (row, feature).
__global__ __launch_bounds__(256) void enf_zsplit_merge_kernel(const float* __restrict__ ysplit, int S, long long BN, int H, int D,
                                                              int N, int Z, int sk_len, float* __restrict__ ybar, float* __restrict__ lse) {
  const int HD = H * D;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BN * HD) return;
  const long long row = e / HD;
  const int c = (int)(e % HD), h = c / D;
  // the parts of this row's query tile: the workgroups whose runs meet its latent steps [tf Z, (tf + 1) Z)
  const int tiles_n = (N + 127) / 128, tf = (int)(row / N) * tiles_n + (int)(row % N) / 128;
  const int parts = (tf * Z + Z - 1) / sk_len - (tf * Z) / sk_len + 1;
  const float* st = ysplit + (size_t)S * BN * HD;
  float ms = -INFINITY;
  for (int s = 0; s < parts; ++s) ms = fmaxf(ms, st[((size_t)s * BN + row) * (H * 3) + h * 3]);
  float L = 0.f, C = 0.f, y = 0.f;
  for (int s = 0; s < parts; ++s) {
    const float* q = st + ((size_t)s * BN + row) * (H * 3) + h * 3;
    const float aw = __expf(q[0] - ms);
    L = fmaf(aw, q[1], L);
    C = fmaf(aw, q[2], C);
    y = fmaf(aw, ysplit[((size_t)s * BN + row) * HD + c], y);
  }
  ybar[(size_t)row * HD + c] = (y - C) / L;
  if (c % D == 0) lse[(size_t)row * H + h] = ms + __logf(L);
}

template <int D, int H, bool BF16, bool ZFOLD, bool MASKS = false, int INV = -1>
static int launch_pair_fwd(const PairFwdArgs& A, hipStream_t st) {
  if constexpr (!MASKS) {
    if (A.mask_mode) return launch_pair_fwd<D, H, BF16, ZFOLD, true>(A, st);       // (the masked passes keep the run-time invariant)
  }
  constexpr int NW = PairWaves<ZFOLD>::NW;
  constexpr bool A3 = ZFOLD && BF16 && ENF_K2_A3 != 0 && NW == 8 && PairCfg<D, BF16>::DD::SPP == 1;
  using SM = PairSmem<D, H, BF16, NW, A3 ? 3 : 2>;
  auto kern = enf_pair_fwd_kernel<D, H, BF16, ZFOLD, MASKS, INV>;
  static EnfAttrBits attr_done{0};          // one per instantiation, one bit per device
  if (!enf_lds_attr(reinterpret_cast<const void*>(kern), SM::TOTAL, attr_done)) return ENF_ELAUNCH;
  dim3 grid((A.N + 16 * A.qg - 1) / (16 * A.qg), A.B);
  if (ZFOLD && A.zsplit > 1) grid = dim3((unsigned)(((long long)grid.x * A.B * A.Z + A.sk_len - 1) / A.sk_len));
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), SM::TOTAL, st, A);
  if (ZFOLD && A.zsplit > 1) {
    const long long BN = (long long)A.B * A.N, tot = BN * H * D;
    hipLaunchKernelGGL(enf_zsplit_merge_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)A.ysplit, A.zsplit, BN, H, D,
                       A.N, A.Z, A.sk_len, A.ybar, A.lse);
  }
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

extern "C" int enf_launch_wz(const EnfDims&, const EnfLayout&, const char*, const float*, char*, float*, char*, char*, hipStream_t);

// wz / wzb: scratch for the z-fold variant (enf_workspace: W.wz, W.wzb), or NULL for the latent-split variant
// relu masks: per call (EnfDims.masks / mask_mode / mask_B, from the descriptor)

extern "C" int enf_launch_pair_fwd(const EnfDims& m, const EnfLayout& L, const char* blob, const float* x, long long x_bstride,
                                   const float* lt, float* ybar, float* lse, char* wz, float* wzb, char* wzu, float* ysplit,
                                   int run_fold, int run_pair, hipStream_t st) {
  PairFwdArgs A;
  A.x = x; A.x_bstride = x_bstride; A.lt = lt; A.blob = blob; A.L = L; A.ybar = ybar; A.lse = lse; A.wz = wz; A.wzb = wzb; A.wzu = wzu; A.inv_d = 1.0f / (float)m.Dt;
  A.B = m.B; A.N = m.N; A.Z = m.Z; A.dx = m.dx; A.inv = m.inv; A.use_window = m.use_window;
  A.masks = m.masks; A.mask_mode = (run_pair & 1) ? m.mask_mode : 0; A.mask_B = m.mask_B;
  // as many latent splits as there are latents to split (up to 8); the rest of the 8 waves take more queries
  int zs = 1;
  while (zs < NWAVES && zs * 2 <= m.Z) zs *= 2;
  A.qg = NWAVES / zs;
  const bool zfold = wz && wzb && wzu && (size_t)m.Z * m.H * enf_panel_bytes(m.D, m.D, m.bf16) < 0x7fffffffu;
  const EnfStreamK sk = enf_zfold_streamk(m);
  A.zsplit = zfold && ysplit && sk.parts > 1 ? sk.parts : 1;
  A.sk_len = A.zsplit > 1 ? sk.len : 0;
  A.ysplit = ysplit;
  A.ybar_half = (run_pair & 2) && A.zsplit == 1 && m.bf16;
  A.xcd_remap = zfold && m.B % 8 == 0 && A.zsplit == 1;
  if (zfold) {
    A.qg = PairWaves<true>::NW;
    if (run_fold) {
      int rc = enf_launch_wz(m, L, blob, lt, wz, wzb, wzu, nullptr, st);
      if (rc) return rc;
    }
  }
  if (!(run_pair & 1)) return 0;
#define ENF_CASE(DD, HH)                                                                                      \
  if (m.D == DD && m.H == HH) {                                                                               \
    if (zfold) return m.bf16 ? launch_pair_fwd<DD, HH, true, true>(A, st) : launch_pair_fwd<DD, HH, false, true>(A, st); \
    return m.bf16 ? launch_pair_fwd<DD, HH, true, false>(A, st) : launch_pair_fwd<DD, HH, false, false>(A, st);         \
  }
#if ENF_K2_INV_SPECIALISED
  if (m.bf16 && m.dx == 2 && !A.mask_mode) {       // the shipped configs' bf16 kernels with the invariant fixed at compile time
    if (m.D == 128 && m.H == 2) {
#define ENF_SPEC(INVID)                                                                                                  \
      if (m.inv == INVID) return zfold ? launch_pair_fwd<128, 2, true, true, false, INVID>(A, st)                          \
                                       : launch_pair_fwd<128, 2, true, false, false, INVID>(A, st);
      ENF_SPEC(ENF_INV_REL_POS_PERIODIC) ENF_SPEC(ENF_INV_LATITUDE_PERIODIC) ENF_SPEC(ENF_INV_POLAR_PERIODIC)
#undef ENF_SPEC
    }
    if (m.D == 64 && m.H == 2 && m.inv == ENF_INV_PONITA)
      return zfold ? launch_pair_fwd<64, 2, true, true, false, ENF_INV_PONITA>(A, st) : launch_pair_fwd<64, 2, true, false, false, ENF_INV_PONITA>(A, st);
  }
#endif
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
  ENF_CASE(64, 4)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
