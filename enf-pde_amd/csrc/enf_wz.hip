// enf_wz.hip -- K1b: per-latent fold of FiLM + the mixer's first Dense (z-fold forward path).
//
// Between the gamma/beta projection and the mixer's gelu there is no nonlinearity (ECA:112-122, ECA:17):
//   v_h   = v0_zh (1 + gamma_h) + beta_h,   [gamma_h | beta_h] = n AGB_h + b            (FiLM, ECA:115-121)
//   a5_h  = v_h AM + bm                                                                    (mixer Dense_0, ECA:17)
// so for one latent z and head h
//   a5_h  = n W_zh + c_zh,   W_zh = (Wgamma_h diag(v0_zh) + Wbeta_h) AM,
//                            c_zh = (v0_zh (1 + bgamma_h) + bbeta_h) AM + bm
// which turns two per-pair GEMMs (D x 2D and D x D) and the FiLM arithmetic into ONE D x D GEMM whose
// matrix depends on the latent only.  This kernel builds W_zh directly in the pair kernel's packed
// A-fragment order and c_zh as fp32 vectors.  2 D^3 FLOP per (latent, head): 8.6 GFLOP for 1024 latents,
// against 2 TFLOP of per-pair work it feeds.
//
// One wave = one (head, 32-row block of W) for a strided set of latents: everything that does not depend on the latent --
// the Wgamma rows, the Wbeta AM tiles the accumulators start from -- is loaded ONCE into registers and reused (the first
// version re-read them per latent: 57 KB of L2 traffic per (latent, head, block) task, 470 MB per call, which bound it).
// The product runs "flipped" (gemm_tile_flip): rows of
// W (the GEMM's input features i) play the role of activation columns, so the accumulator tile of
// (i-tile, k-tile) holds, lane by lane, exactly the elements of that lane's A-fragment (k-tile, i-block):
// the result is converted and stored 16 bytes per lane, lane-linear, no transpose.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "enf_layout.h"
#include "enf_launch.h"
#include "enf_device.h"

struct WzArgs {
  const float* lt; const char* blob; EnfLayout L;
  char* wz; float* wzb; char* wzu;
  char* wzt;             // backward-orientation panels (NULL: not wanted)
  size_t pstride;        // bytes between the panels of consecutive (latent, head) pairs (wz and wzt alike)
  int BZ;
};

constexpr int WZ_WAVES = 4;
#ifndef ENF_WZ_BWD_GRID
#define ENF_WZ_BWD_GRID 128     // the backward's call (both orientations) runs beside the tail kernel of an inner step (64 workgroups)
#endif
#ifndef ENF_WZ_MAXGRID
#define ENF_WZ_MAXGRID 256      // x 4 waves = one per SIMD (the bf16 128-wide instantiation holds 352 registers)
#endif

template <int D, int H, bool BF16>
__global__ __launch_bounds__(64 * WZ_WAVES) void enf_wz_kernel(WzArgs A) {
  constexpr int KB = D / 32, NT = D / 16;
  constexpr int PB = D * D * (BF16 ? 2 : 4);                      // bytes of one packed D x D panel
  extern __shared__ __attribute__((aligned(16))) char smem[];     // the AM forward panel (B operand here)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, quad = lane >> 4;
  {
    const char* src = A.blob + A.L.am;
    for (int o = tid * 16; o < PB; o += 64 * WZ_WAVES * 16)
      *reinterpret_cast<f32x4*>(smem + o) = *reinterpret_cast<const f32x4*>(src + o);
  }
  __syncthreads();
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(A.blob + off); };
  const float* agb = G(A.L.p_agb);
  const int ltstride = enf_lt_stride(H, D);
  // (gridDim.x * WZ_WAVES is a multiple of H * KB: launch_wz)
  constexpr int COMBOS = H * KB;
  const int gw = blockIdx.x * WZ_WAVES + wave, lat_stride = gridDim.x * WZ_WAVES / COMBOS;
  const int blk = (gw % COMBOS) % KB, h = (gw % COMBOS) / KB;
  // latent-independent operands of this (head, block), resident for the whole sweep
  f32x4 wg[2][NT], bf[2][NT];
  const float* wbmt = G(A.L.p_wbmt) + (size_t)h * D * D;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float* row = agb + (size_t)(16 * (2 * blk + a) + col) * (2 * H * D) + h * 2 * D;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      wg[a][tj] = *reinterpret_cast<const f32x4*>(row + 64 * (tj >> 1) + 16 * (tj & 1) + 4 * quad);
      bf[a][tj] = *reinterpret_cast<const f32x4*>(wbmt + (size_t)(16 * tj + col) * D + 16 * (2 * blk + a) + 4 * quad);
    }
  }
  f32x4 bb[2][NT];                                               // backward orientation: Wbeta AM tiles, un-flipped
  if (A.wzt) {
    const float* wbm = G(A.L.p_wbm) + (size_t)h * D * D;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
        bb[a][kt] = *reinterpret_cast<const f32x4*>(wbm + (size_t)(16 * (2 * blk + a) + col) * D + 16 * kt + 4 * quad);
  }
  // bias row c_zh: every wave takes NT / KB of its k-tiles (one wave doing all of them was the slowest of its group);
  // its latent-independent operands are resident too
  constexpr int BT = NT / KB;
  f32x4 gq[NT];
  float cbv[BT];
  {
    const float* opbg = G(A.L.p_opbg) + h * D;
    const float* cb = G(A.L.p_cb) + h * D;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) gq[tj] = *reinterpret_cast<const f32x4*>(opbg + 16 * tj + 4 * quad);
#pragma unroll
    for (int t = 0; t < BT; ++t) cbv[t] = cb[16 * (BT * blk + t) + col];
  }
  for (int bz = gw / COMBOS; bz < A.BZ; bz += lat_stride) {
    const float* v0 = A.lt + (size_t)bz * ltstride + enf_lt_off_v0(H, D) + h * D;
    // "activation" fragments: X[i][j] = Wgamma_h[i][j] v0[j], rows i = 16 (2 blk + a) + col as columns
    Frags<BF16, KB> FX[2];
    {
      f32x4 v[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) v[tj] = *reinterpret_cast<const f32x4*>(v0 + 16 * tj + 4 * quad);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x4 X[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) X[tj] = wg[a][tj] * v[tj];
        make_frags<BF16, KB>(FX[a], X);
      }
      // bias row: column 0 carries v0 (1 + bgamma_h); the product's row 0 is c_zh - cb_h
      f32x4 X[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) X[tj] = col == 0 ? v[tj] * gq[tj] : f32x4{0.f, 0.f, 0.f, 0.f};
      Frags<BF16, KB> FB;
      make_frags<BF16, KB>(FB, X);
      float* dst = A.wzb + (size_t)(bz * H + h) * D;
#pragma unroll
      for (int t = 0; t < BT; ++t) {
        const int kt = BT * blk + t;
        f32x4 ab = {0.f, 0.f, 0.f, 0.f};
        gemm_tile_flip<BF16, KB>(ab, FB, smem, kt, lane);
        if (quad == 0) dst[16 * kt + col] = ab[0] + cbv[t];
      }
    }
    char* panel = A.wz + (size_t)(bz * H + h) * A.pstride;
    {
      // forward orientation: flipped product of all NT k-tiles for the two i-tiles of this block (hand-scheduled
      // stage when available), then lane-linear fragment stores
      f32x4 af[2][NT];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) af[a][kt] = bf[a][kt];
        if constexpr (BF16 && ENF_ASM_GEMM && GemmStageAsm<KB, NT>::available) {
          GemmStageAsm<KB, NT>::run_flip(af[a], FX[a].f, (unsigned)(uintptr_t)(lds_ptr_t)(smem + (lane << 4)));
        } else {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt) gemm_tile_flip<BF16, KB>(af[a][kt], FX[a], smem, kt, lane);
        }
      }
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if constexpr (BF16) {
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (__bf16)af[j >> 2][kt][j & 3];
          *reinterpret_cast<bf16x8*>(panel + (((kt * KB + blk) * 64 + lane) << 4)) = o;
        } else {
#pragma unroll
          for (int a = 0; a < 2; ++a)
            *reinterpret_cast<f32x4*>(panel + (((kt * 2 * KB + 2 * blk + a) * 64 + lane) << 4)) = af[a][kt];
        }
      }
    }
    if (A.wzt) {
      // the same matrix in backward orientation (A[i][k] = W_zh[i][k], rows = the forward GEMM's INPUT index):
      // the un-flipped product puts (k-tile, i-columns) tiles in the accumulators, whose lanes hold exactly the
      // elements of fragment (mt = i-tile, blk = k-block)
      char* panel_t = A.wzt + (size_t)(bz * H + h) * A.pstride;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int it = 2 * blk + a;
        f32x4 acc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) acc[kt] = bb[a][kt];
        gemm_stage<BF16, KB, NT>(acc, FX[a], smem, lane);
        if constexpr (BF16) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)acc[2 * kb + (j >> 2)][j & 3];
            *reinterpret_cast<bf16x8*>(panel_t + (((it * KB + kb) * 64 + lane) << 4)) = o;
          }
        } else {
#pragma unroll
          for (int kt = 0; kt < NT; ++kt)
            *reinterpret_cast<f32x4*>(panel_t + (((it * 2 * KB + kt) * 64 + lane) << 4)) = acc[kt];
        }
      }
    }
    if (BF16 && A.wzu && blk == 0 && h == 0) {
      // logit vectors as the rows of a bf16 A operand: entry ((kb*4 + kq)*H + hh) = the 8 k-values lane
      // (m = hh, kq) of block kb feeds v_mfma_f32_16x16x32_bf16 (rows m >= H are zero and not stored)
      if (lane < KB * 4 * H) {
        const int hh = lane % H, kq = (lane / H) % 4, kb = lane / (4 * H);
        const float* u = A.lt + (size_t)bz * ltstride + enf_lt_off_u(H, D) + hh * D + 32 * kb + 4 * kq;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(u), hi = *reinterpret_cast<const f32x4*>(u + 16);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = (__bf16)lo[j]; o[4 + j] = (__bf16)hi[j]; }
        *reinterpret_cast<bf16x8*>(A.wzu + (size_t)bz * enf_wzu_bytes(H, D) + lane * 16) = o;
      }
    }
  }
}

template <int D, int H, bool BF16>
static int launch_wz(const WzArgs& A, hipStream_t st) {
  constexpr int PB = D * D * (BF16 ? 2 : 4);
  auto kern = enf_wz_kernel<D, H, BF16>;
  static EnfAttrBits attr_done{0};          // one per instantiation, one bit per device
  if (!enf_lds_attr(reinterpret_cast<const void*>(kern), PB, attr_done)) return ENF_ELAUNCH;
  // waves = COMBOS (head, block) roles x a number of latent lanes; each wave sweeps BZ / lanes latents
  constexpr int COMBOS = H * (D / 32);
  // the backward's call (both orientations) runs on the side stream beside the tail kernels and is off the critical
  // path with half the chip (same-box A/B of the fit: 3.036 ms with 128 workgroups, 3.052 with 256); the decode's call
  // is ON the critical path and takes the whole chip
#ifdef ENF_AB_SWITCHES       // A/B builds only: ENF_WZ_GRID=n overrides the workgroup count
  static int envgrid = -1;
  if (envgrid < 0) { const char* e = getenv("ENF_WZ_GRID"); envgrid = e ? atoi(e) : 0; if (envgrid == 1) envgrid = 2; }
#else
  constexpr int envgrid = 0;
#endif
  const int maxgrid = envgrid > 0 ? envgrid : (A.wzt ? ENF_WZ_BWD_GRID : ENF_WZ_MAXGRID);
  const int max_waves = maxgrid * WZ_WAVES;
  int lanes = A.BZ;
  while (lanes * COMBOS > max_waves && lanes > 1) lanes = (lanes + 1) / 2;
  int grid = (lanes * COMBOS + WZ_WAVES - 1) / WZ_WAVES;
  while ((grid * WZ_WAVES) % COMBOS) ++grid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WZ_WAVES), PB, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

extern "C" int enf_launch_wz(const EnfDims& m, const EnfLayout& L, const char* blob, const float* lt, char* wz, float* wzb,
                             char* wzu, char* wzt, hipStream_t st) {
  // wzt == NULL: forward panels only, packed back to back in wz (the z-fold forward kernel's layout);
  // wzt != NULL: `wzt` holds [forward | backward] panel pairs per (latent, head) and `wz` is ignored
  WzArgs A;
  const size_t PB = enf_panel_bytes(m.D, m.D, m.bf16);
  A.lt = lt; A.blob = blob; A.L = L; A.wzb = wzb; A.wzu = wzu; A.BZ = m.B * m.Z;
  if (wzt) { A.wz = wzt; A.wzt = wzt + PB; A.pstride = 2 * PB; }
  else { A.wz = wz; A.wzt = nullptr; A.pstride = PB; }
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) return m.bf16 ? launch_wz<DD, HH, true>(A, st) : launch_wz<DD, HH, false>(A, st);
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
  ENF_CASE(64, 4)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
