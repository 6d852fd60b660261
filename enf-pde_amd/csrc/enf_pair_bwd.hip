// enf_pair_bwd.hip -- K3: gradient of the pair chain w.r.t. the per-latent quantities
// (u, v0, c, pose, window coefficient), i.e. the part of jax.grad(loss)(latents)
// (pde_trainer.py:188,200) that touches every (query, latent) pair.
//
// "z-major": one wave owns ONE latent z of one signal and sweeps tiles of 16 queries
// (columns = queries).  Everything per pair is recomputed in registers (nothing of size
// B*N*Z is ever stored); reductions over queries -- the gradient of a per-latent quantity --
// stay lane-local across the sweep and are folded across lanes once, at the end.
// Softmax backward needs no reduction over z: with the forward's log-sum-exp and
// delta[n,h] = d(ybar)[n,h,:] . ybar[n,h,:] (computed by the tail backward),
//     att = exp(logit - lse),  d(logit) = att * (d(ybar).n~ - delta).
// Per tile (H heads):  q-forward (logits) -> v-forward to n^ -> per head {gamma/beta, mixer
// forward, mixer backward, FiLM backward, gamma/beta backward} -> LN/gelu/relu backward to the
// value RFF -> q-branch recompute + backward -> invariant Jacobian.  608 MFMAs (16x16x32 bf16)
// per tile at D=128, H=2 against 288 in the forward.  The two big per-latent sums
// (d u = sum_n dlogit h1, d v0 = sum_n dv (1+gamma)) are taken on FLIPPED products (enf_device.h:
// gemm_tile_flip -- same panels, MFMA operands swapped) whose rows are the queries, so they cost 4
// FMAs per 16x16 tile and 32 accumulator registers instead of 128; +160 MFMAs per tile.
#include <hip/hip_runtime.h>
#ifndef ENF_K3_LITE
#define ENF_K3_LITE 0
#endif
#define ENF_ASM_LITE ENF_K3_LITE
#include "enf_layout.h"
#include "enf_launch.h"
#include "enf_device.h"
#include "enf_pair_common.h"

#ifndef ENF_K3_FENCE
#define ENF_K3_FENCE 1
#endif
#if ENF_K3_FENCE
#define K3_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define K3_SCHED_FENCE() do {} while (0)
#endif
#ifndef ENF_K3_OPAQUE
#define ENF_K3_OPAQUE 1
#endif
#if ENF_K3_OPAQUE
#define K3_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define K3_OPAQUE(x) do {} while (0)
#endif
#ifndef ENF_K3_EARLY_DY
#define ENF_K3_EARLY_DY 0
#endif
#ifndef ENF_K3_PARK
#define ENF_K3_PARK 1
#endif
#ifndef ENF_K3_FUSED_GELU
#define ENF_K3_FUSED_GELU 0
#endif
#ifndef ENF_K3_ZF_FUSED       // z-fold heads: gelu(a5) and gelu'(a5) from one exp + rcp (a5 is overwritten by its gelu')
#define ENF_K3_ZF_FUSED 1
#endif
#ifndef ENF_K3_ENTRY_BARRIER  // round 1's timing mitigation of the run-to-run deviations (a barrier as the kernel's first statement).  OFF since
#define ENF_K3_ENTRY_BARRIER 0  // the cause is fixed (enf_device.h: ln_apply): probes and suite clean without it (scripts/k3_race/README.md)
#endif
#ifndef ENF_K3_ANTI           // the upper four waves (the SIMD-mates of the lower four) take each stage's barrier BEFORE its MFMAs
#define ENF_K3_ANTI 0         // (enf_device.h: Pipe.early): one wave of a SIMD multiplies while the other runs its vector epilogue.
#endif                        // OFF: correct (148 tests) but 9 % SLOWER on the fit (3.00 vs 2.75 ms, gpurun_out/r02/ab_anti.log), as in K2
#ifndef ENF_K3_PREFETCH       // per-tile global reads (query coordinates, lse) issued one tile ahead.  OFF: measured 1.4 % SLOWER
#define ENF_K3_PREFETCH 0     // on the fit (2.74 -> 2.78 ms same-box, gpurun_out/r02/ab_pf.log): the five live registers cost more
#endif                        // than the exposed L2 latency at the top of a tile; with delta prefetched too, +25 spilled dwords
#if ENF_K3_PREFETCH
#define K3_LSE(h) t_lse[h]
#else
#define K3_LSE(h) A.lse[qrow * H + (h)]
#endif
#define K3_DELTA(h) A.delta[qrow * H + (h)]     // (delta is wanted late in the tile: prefetching it costs 25 spilled dwords)
#ifndef ENF_STORE_NT          // activation store (STORE instantiation): non-temporal stores
#define ENF_STORE_NT 0
#endif
#ifndef ENF_K3_A3_FUSED       // value chain: park gelu'(a3) (from the sigmoid gelu(a3) needs anyway) instead of a3
#define ENF_K3_A3_FUSED 1
#endif
#ifndef ENF_K3_UF_FUSED       // unfolded heads: the same
#define ENF_K3_UF_FUSED 1
#endif
#ifndef ENF_K3_LA             // z-fold bf16: look-ahead staging (enf_device.h: panel_gemm<.., LA>) -- one stage always in flight
#define ENF_K3_LA 1
#endif
#ifndef ENF_K3_LDSACC
#define ENF_K3_LDSACC 1
#endif
#ifndef ENF_K3_ZF_SPLIT       // z-fold at a full chip: workgroups per latent (query tiles split), co-located on one XCD (launch code below)
#define ENF_K3_ZF_SPLIT 1
#endif
#ifndef ENF_K3_DN_LATE        // z-fold: the d n^ GEMMs run after both heads (their inputs parked as fragments) instead of inside them
#define ENF_K3_DN_LATE 1
#endif
#ifndef ENF_K3_GELU_PK        // the fused gelu / gelu' with its polynomial parts as packed instructions (gelu_fg_tile)
#define ENF_K3_GELU_PK 1
#endif
#ifndef ENF_K3_DOT_PK         // the four-term dot products of the flipped d v0 sums as packed multiplies
#define ENF_K3_DOT_PK 1
#endif
#ifndef ENF_K3_ZF_EARLY_DY    // z-fold heads: d ybar / delta requested before the head's first GEMM stage
#define ENF_K3_ZF_EARLY_DY 1
#endif
#ifndef ENF_K3_INV_SPECIALISED
#define ENF_K3_INV_SPECIALISED 1
#endif
#ifndef ENF_K3_STORE_SPEC
#define ENF_K3_STORE_SPEC 1
#endif
#ifndef ENF_K3_STATIC_PRIO
#define ENF_K3_STATIC_PRIO 0
#endif
#ifndef ENF_K3_XCD_REMAP
#define ENF_K3_XCD_REMAP 1
#endif

struct PairBwdArgs {
  const float* x; long long x_bstride;
  const float* lt; const char* blob; EnfLayout L;
  const float* lse; const float* dybar; const float* delta;
  float* dlt;
  void* store[ENF_NUM_STORE(4)];        // ENF_S_* buffers (STORE instantiation only)
  float inv_d;                          // 1 / (true num_hidden)
  const char* wzt; const float* wzb;    // ZF only: per (latent, head) [forward | backward] panels of W_zh, and c_zh
  float* dxq;                           // (B, N, dx) or nullptr: gradient w.r.t. the query coordinates, accumulated (atomics)
  const unsigned* masks; int mask_B;    // STORE only: relu masks to linearise at (ENF_MASK_READ), or nullptr
  int mask_b0;                          // signal index of this launch's b = 0 in the caller's batch (chunked weight-gradient passes)
  int B, N, Z, dx, inv, use_window, nsplit;
  int xcd_remap;                        // ZF: 1-D grid, the nsplit workgroups of a latent adjacent on one XCD (launch_pair_bwd)
};

// The STORE instantiation's 7 + 4 H buffer pointers, fetched from the kernel-argument segment where they are used (s_load_dwordx2 +
// lgkmcnt).  Read as `A.store[i]` hipcc loads all of them once, cannot keep 30 scalar registers across the tile loop and spills them to
// scratch: 15 scratch reloads per tile, each with an `s_waitcnt vmcnt(0)` that also waits for the LDS-DMA stage in flight.
#ifndef ENF_K3_STORE_SLOAD
#define ENF_K3_STORE_SLOAD 1
#endif
DEV void* k3_store_ptr(const PairBwdArgs& A, int i) {
#if ENF_K3_STORE_SLOAD
  void* p;
  const unsigned off = (unsigned)(offsetof(PairBwdArgs, store) + 8 * i);
  asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(p) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "s"(off));
  return p;
#else
  return A.store[i];
#endif
}

// Debug build only (-DENF_STAMPS): s_memtime stamps of the first tiles of one workgroup (scripts/stamps_k3.py)
#ifdef ENF_STAMPS
__device__ unsigned long long enf_stamps_bwd[8 * 4 * 24];
#define BSTAMP(k)                                                                                             \
  do {                                                                                                        \
    if (blockIdx.x == 7 && blockIdx.y == 0 && lane == 0 && ti < 4)                                            \
      enf_stamps_bwd[(wave * 4 + ti) * 24 + (k)] = __builtin_amdgcn_s_memtime();                              \
  } while (0)
extern "C" int enf_debug_read_stamps_bwd(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(enf_stamps_bwd), sizeof(enf_stamps_bwd)) == hipSuccess ? 0 : -1;
}
// workgroup-level stamps (kernel entry, tile loop start / end, exit) of workgroups 7, 263, 519, 775 (one per round at 256 CUs)
__device__ unsigned long long enf_stamps_bwd_wg[4 * 8 * 4];
#define WSTAMP(k)                                                                                             \
  do {                                                                                                        \
    if ((blockIdx.x & 255) == 7 && blockIdx.x < 1024 && blockIdx.y == 0 && lane == 0)                         \
      enf_stamps_bwd_wg[((blockIdx.x >> 8) * 8 + wave) * 4 + (k)] = __builtin_amdgcn_s_memtime();             \
  } while (0)
extern "C" int enf_debug_read_stamps_bwd_wg(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(enf_stamps_bwd_wg), sizeof(enf_stamps_bwd_wg)) == hipSuccess ? 0 : -1;
}
#else
#define BSTAMP(k) do {} while (0)
#define WSTAMP(k) do {} while (0)
#endif

#ifdef ENF_TEST_HOOKS
// Test-only build (libenf_hip_test.so, `make test-lib`): the kernel's EPILOGUE also writes every wave's final per-latent sums,
// so a test can compare the waves that recompute the same latent (the inactive waves of the last workgroup keep the barrier
// cadence on the last latent) bit for bit inside ONE launch.  The tile loop is untouched; the product library has none of this.
constexpr int ENF_HOOK_WGS = 64, ENF_HOOK_ROW = 16 * 64 + 16;
__device__ float enf_hook_wave_sums[ENF_HOOK_WGS * NWAVES * ENF_HOOK_ROW];
extern "C" int enf_test_read_wave_sums(float* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(enf_hook_wave_sums), sizeof(enf_hook_wave_sums)) == hipSuccess ? 0 : -1;
}
#endif

// one row of a materialised activation / delta = this lane's share of a fragment set.
// bf16: the 8 values of a fragment go out as ONE 16-byte store at columns 32 blk + 8 quad + j, i.e. the row is stored
// with its columns PERMUTED inside every 32-block (true feature = 32 blk + 4 quad + j for j < 4, 32 blk + 16 + 4 quad +
// j - 4 otherwise; ENF_S_* in include/enf_hip.h).  Every buffer uses the same permutation, the consumers are products
// X^T delta over the pair axis, so the caller un-permutes the small D x D results instead.  Half the store instructions
// and 64-byte instead of 32-byte row segments: the store is what bounds this instantiation.
template <bool BF16, int KB>
DEV void store_frags(void* base, size_t row, int D, const Frags<BF16, KB>& F, int quad) {
  if constexpr (BF16) {
    __bf16* p = reinterpret_cast<__bf16*>(base) + row * D;
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
#if ENF_STORE_NT
      // written once, read once by enf_xtd_kernel, 2 GB per pass: streaming stores keep it out of the way of the panels in L2
      __builtin_nontemporal_store(__builtin_bit_cast(f32x4, F.f[blk]), reinterpret_cast<f32x4*>(p + 32 * blk + 8 * quad));
#else
      *reinterpret_cast<bf16x8*>(p + 32 * blk + 8 * quad) = F.f[blk];
#endif
    }
  } else {
    float* p = reinterpret_cast<float*>(base) + row * D;
#pragma unroll
    for (int t = 0; t < 2 * KB; ++t) *reinterpret_cast<f32x4*>(p + 16 * t + 4 * quad) = F.f[t];
  }
}

template <int D, int H, bool BF16> struct PairBwdSmem {
  static constexpr int RING = 0;
  static constexpr int CONSTS = RING + 2 * STAGE_MAX;
  // bq1 bv1 bf bm (D each) | bgb (2HD) | acq acv (2D each)
  static constexpr int N_CONST = 4 * D + 2 * H * D + 4 * D;
  static constexpr int GC = CONSTS + 4 * N_CONST;                          // gcq | gcv panels
  static constexpr int GC_BYTES = PanelCfg<D / 64, 1, BF16>::BYTES;
  static constexpr int ZVEC = GC + 2 * GC_BYTES;                           // NWAVES x 2HD floats
  static constexpr int LACC = ZVEC + 4 * NWAVES * 2 * H * D;               // NWAVES x (2 H D/16) x 64 lanes floats: dU | dV0 partial sums
  // ball / ball_lat: NWAVES x 11 x 16 floats, per-column partial sums of d R (9, quad-0 lanes) and of the two latent-only
  // invariants' gradients (quad-1 lanes)
  static constexpr int EXTACC = LACC + (ENF_K3_LDSACC ? 4 * NWAVES * 2 * H * (D / 16) * 64 : 0);
  static constexpr bool EXT_OK = D == 64;            // the 128-wide kernels have no LDS left: ball needs D = 64 (enf_check_desc)
  static constexpr int TOTAL = EXTACC + (EXT_OK ? 4 * NWAVES * 11 * 16 : 0);
  static_assert(TOTAL <= 160 * 1024, "K3 LDS budget");
};

// d t = d E_sin * E_cos - d E_cos * E_sin   (the 2 pi is folded into the gc panel)
template <int D> DEV void rff_embed_bwd(f32x4 (&dT)[D / 32], const f32x4 (&dE)[D / 16], const f32x4 (&E)[D / 16]) {
  constexpr int TT = D / 32;
#pragma unroll
  for (int m = 0; m < TT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) dT[m][i] = dE[m][i] * E[TT + m][i] - dE[TT + m][i] * E[m][i];
}

// gamma/beta panel of one head: transposed product -> v = v0 (1+gamma) + beta (as gb_panel), and the
// FLIPPED product of the gamma tiles -> opgf[d-tile] = 1 + gamma with rows = queries (for d v0).
template <int D, bool BF16, int NEXT_BYTES>
DEV void gb_panel_flip(f32x4 (&v)[D / 16], f32x4 (&opgf)[D / 16], const Frags<BF16, D / 32>& F, Pipe& P, char* ring,
                       unsigned panel, unsigned next, const float* bias, const float* v0vec, int lane, int col, int quad) {
  using C = typename PairCfg<D, BF16>::GB;
  constexpr int KB = D / 32, MTS = C::MTS;
#pragma unroll
  for (int sp = 0; sp < C::SPP; ++sp) {
    if (sp + 1 < C::SPP) stage_issue<C::STAGE>(P.rs, panel + (sp + 1) * C::STAGE, ring + (P.cur ^ 1) * STAGE_MAX, P.wave, lane);
    else if (next != NO_STAGE) stage_issue<NEXT_BYTES>(P.rs, next, ring + (P.cur ^ 1) * STAGE_MAX, P.wave, lane);
    const char* slot = ring + P.cur * STAGE_MAX;
    f32x4 t[MTS];
    constexpr bool ASM = BF16 && ENF_ASM_GEMM && GemmStageAsm<KB, MTS, ENF_ASM_LITE != 0>::available && MTS == 8;
    if constexpr (ASM) {
      // transposed product of all 8 tiles + flipped product of the 4 gamma tiles {0,1,4,5} from one fragment read
      f32x4 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bc = 1.0f + bias[16 * (sp * MTS + 4 * (j >> 1) + (j & 1)) + col];
        af[j] = f32x4{bc, bc, bc, bc};
      }
      GemmStageAsm<KB, MTS, ENF_ASM_LITE != 0>::run_gb_bias(t, af, F.f, (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<char*>(slot) + (lane << 4)),
                                                            (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<float*>(bias) + 16 * sp * MTS + 4 * quad));
#pragma unroll
      for (int j = 0; j < 4; ++j) opgf[2 * (sp * (MTS / 4) + (j >> 1)) + (j & 1)] = af[j];
    } else {
#pragma unroll
      for (int j = 0; j < MTS; ++j) t[j] = rowvec(bias, sp * MTS + j, quad);
      gemm_stage<BF16, KB, MTS>(t, F, slot, lane);
    }
#pragma unroll
    for (int g = 0; g < MTS / 4; ++g) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int tile = 2 * (sp * (MTS / 4) + g) + e;
        const f32x4 v0 = rowvec(v0vec, tile, quad);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[tile][i] = fmaf(v0[i], 1.0f + t[4 * g + e][i], t[4 * g + 2 + e][i]);
        if constexpr (!ASM) {
          const float bc = 1.0f + bias[16 * (sp * MTS + 4 * g + e) + col];      // per-column bias of the flipped tile
          f32x4 af = {bc, bc, bc, bc};
          gemm_tile_flip<BF16, KB>(af, F, slot, 4 * g + e, lane);
          opgf[tile] = af;
        }
      }
    }
    stage_wait();
    __syncthreads();
    P.cur ^= 1;
  }
}

// Jacobian of (invariant, window) w.r.t. the latent pose row and the window coefficient.
// dinv: gradient of the I invariant components; dwin: gradient of the additive window term.
template <bool FAST>
DEV void pair_invariant_bwd(int inv_id, int dx, const QueryPt& q, const f32x4& pz, float wcoef, int use_window,
                            const float (&inv)[4], float win, float (&dinv)[4], float dwin, float (&dpose)[4],
                            float& dwc, float* dR = nullptr, float* dxq = nullptr, const float* ext = nullptr) {
  // dxq (3 values, or nullptr): the same pair's gradient w.r.t. the QUERY coordinates.  Where invariant and window depend
  // on (x - p) only it is minus this pair's pose-position gradient; the spherical ones add their theta_x (and r_x) terms.
  const float PI = 3.14159265358979323846f;
  const float dp0 = dpose[0], dp1 = dpose[1];
  switch (inv_id) {
    case ENF_INV_REL_POS_PERIODIC: {
      if (use_window) {                                   // win = wc (c0^2 + c1^2)
        dwc += dwin * (inv[0] * inv[0] + inv[1] * inv[1]);
        dinv[0] += dwin * 2.f * wcoef * inv[0];
        dinv[1] += dwin * 2.f * wcoef * inv[1];
      }
      dpose[0] += PI * (-inv[2] * dinv[0] + inv[0] * dinv[2]);   // d/dD0 [cos pi D0, sin pi D0], D = p - x
      dpose[1] += PI * (-inv[3] * dinv[1] + inv[1] * dinv[3]);
      if (dxq) { dxq[0] = -(dpose[0] - dp0); dxq[1] = -(dpose[1] - dp1); dxq[2] = 0.f; }
    } break;
    case ENF_INV_BALL:                 // inv = [R x^ (3), r_x]: d R[i][j] += dinv[i] x^[j]   (dR: 9 per-lane sums)
    case ENF_INV_BALL_LAT:             // inv = [th_x, cd, sd, r_x]; th_p and r_p are latent-only rows (the phase)
    case ENF_INV_LATITUDE_PERIODIC:
    case ENF_INV_POLAR_PERIODIC: {
      const float dphi = (q.x0 - pz[0]) * 0.15915494309189535f;
      const float cd = cos_rev<FAST>(dphi), sd = sin_rev<FAST>(dphi);
      const float dot = q.sx * pz[2] * cd + q.cx * pz[3];
      float ddot = 0.f, dcd = 0.f, dsd = 0.f;
      if (inv_id == ENF_INV_LATITUDE_PERIODIC) { dpose[1] += dinv[1]; dcd += dinv[2]; dsd += dinv[3]; }
      else if (inv_id == ENF_INV_BALL_LAT) { dcd += dinv[1]; dsd += dinv[2]; }
      else if (inv_id == ENF_INV_BALL) {
        const float xr = q.x0 * 0.15915494309189535f;
        const float xh[3] = {q.sx * cos_rev<FAST>(xr), q.sx * sin_rev<FAST>(xr), q.cx};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) dR[3 * i + j] = dinv[i] * xh[j];
      }
      else ddot += dinv[0];
      if (use_window) {                                   // win = exp(-ang^2 wc), ang = acos(clip(dot))
        const float dc = fminf(fmaxf(dot, -1.f + 1e-6f), 1.f - 1e-6f);
        const float ang = acosf(dc);
        dwc += dwin * (-ang * ang * win);
        if (dot > -1.f + 1e-6f && dot < 1.f - 1e-6f)
          ddot += dwin * (2.f * ang * wcoef * win) * rsqrtf(1.f - dc * dc);
      }
      dcd += ddot * q.sx * pz[2];
      dpose[2] += ddot * q.sx * cd;                        // d/d sin(theta_p)
      dpose[3] += ddot * q.cx;                             // d/d cos(theta_p)
      dpose[0] += dcd * sd - dsd * cd;                     // d cd/d phi_p = sd, d sd/d phi_p = -cd
      if (dxq) {
        dxq[0] = -(dcd * sd - dsd * cd);                   // phi_x enters through (phi_x - phi_p) ...
        dxq[1] = ddot * (q.cx * pz[2] * cd - q.sx * pz[3]);   // d dot / d theta_x
        dxq[2] = 0.f;
        if (inv_id == ENF_INV_LATITUDE_PERIODIC || inv_id == ENF_INV_BALL_LAT) dxq[1] += dinv[0];   // inv[0] = theta_x
        if (inv_id == ENF_INV_BALL_LAT) dxq[2] = dinv[3];                                           // inv[3] = r_x
        if (inv_id == ENF_INV_BALL) {                      // ... and, for ball, through x^ in R x^: g = R^T dinv
          const float xr = q.x0 * 0.15915494309189535f;
          const float cp = cos_rev<FAST>(xr), sp = sin_rev<FAST>(xr);
          const float g0 = ext[0] * dinv[0] + ext[3] * dinv[1] + ext[6] * dinv[2];
          const float g1 = ext[1] * dinv[0] + ext[4] * dinv[1] + ext[7] * dinv[2];
          const float g2 = ext[2] * dinv[0] + ext[5] * dinv[1] + ext[8] * dinv[2];
          dxq[0] += q.sx * (-g0 * sp + g1 * cp);
          dxq[1] += q.cx * (g0 * cp + g1 * sp) - q.sx * g2;
          dxq[2] = dinv[3];
        }
      }
    } break;
    case ENF_INV_PONITA_FULL:
    case ENF_INV_PONITA: {
      const float r0 = q.x0 - pz[0], r1 = q.x1 - pz[1];
      float dr0 = dinv[0] * pz[2] - dinv[1] * pz[3];
      float dr1 = dinv[0] * pz[3] + dinv[1] * pz[2];
      dpose[2] += dinv[0] * r0 + dinv[1] * r1;
      dpose[3] += dinv[0] * r1 - dinv[1] * r0;
      float dth = 0.f;                                   // d / d theta_x of inv[2] = cos(theta_x) c_p + sin(theta_x) s_p
      if (inv_id == ENF_INV_PONITA_FULL) {
        dpose[2] += dinv[2] * q.cx; dpose[3] += dinv[2] * q.sx;
        dth = dinv[2] * (-q.sx * pz[2] + q.cx * pz[3]);
      }
      if (use_window) {                                   // win = -wc |r|^2
        dwc += dwin * (-(r0 * r0 + r1 * r1));
        dr0 += dwin * (-2.f * wcoef * r0);
        dr1 += dwin * (-2.f * wcoef * r1);
      }
      dpose[0] -= dr0; dpose[1] -= dr1;
      if (dxq) { dxq[0] = dr0; dxq[1] = dr1; dxq[2] = dth; }
    } break;
    default: {
      const float r0 = q.x0 - pz[0], r1 = dx > 1 ? q.x1 - pz[1] : 0.f, r2 = dx > 2 ? q.x2 - pz[2] : 0.f;
      const float d2 = r0 * r0 + r1 * r1 + r2 * r2;
      float dr0 = 0.f, dr1 = 0.f, dr2 = 0.f;
      if (inv_id == ENF_INV_REL_POS) { dr0 = dinv[0]; dr1 = dinv[1]; dr2 = dinv[2]; }
      else if (inv_id == ENF_INV_NORM_REL_POS) {
        const float s = d2 > 0.f ? dinv[0] * rsqrtf(d2) : 0.f;
        dr0 = s * r0; dr1 = s * r1; dr2 = s * r2;
      }
      if (use_window) {
        dwc += dwin * (-d2);
        dr0 += dwin * (-2.f * wcoef * r0); dr1 += dwin * (-2.f * wcoef * r1); dr2 += dwin * (-2.f * wcoef * r2);
      }
      dpose[0] -= dr0;
      if (dx > 1) dpose[1] -= dr1;
      if (dx > 2) dpose[2] -= dr2;
      if (dxq) {
        dxq[0] = dr0; dxq[1] = dr1; dxq[2] = dr2;
        if (inv_id == ENF_INV_ABS_POS) { dxq[0] += dinv[0]; dxq[1] += dinv[1]; dxq[2] += dinv[2]; }   // inv = x
      }
    } break;
  }
}

// ZF (z-fold backward, no STORE): ONE latent per workgroup, the 8 waves take 8 query tiles per sweep step, and per
// head the chain uses the per-latent fold of enf_wz.hip:  a5 = W_zh^T n + c_zh  (one GEMM instead of the gamma/beta
// GEMM + FiLM + mixer Dense),  d n += W_zh d a5  (one GEMM instead of AM^T and AGB^T), and for d v0 the two flipped
// products  dv = (AM d a5)^T,  1 + gamma = (Wgamma_h^T n)^T + (1 + bgamma):  512 MFMAs per tile instead of 788.
// INV >= 0: the invariant is a compile-time constant (the instantiations the shipped configs run: launch code below).  With the
// invariant chosen at run time the tile loop carries the switch of pair_invariant / pair_invariant_bwd -- ~2000 instructions of
// wave-uniform branches whose join points cost spilled registers, and every scratch reload's s_waitcnt vmcnt(0) also waits for
// the LDS-DMA stage in flight: 4.9 k of a tile's 66 k cycles sat between the last GEMM stage and the next tile's first one.
template <int D, int H, bool BF16, bool STORE, bool ZF, int INV = -1>
__global__ __launch_bounds__(NTHREADS, 2) void enf_pair_bwd_kernel(PairBwdArgs A) {
  static_assert(!(ZF && STORE), "the activation store needs the unfolded chain");
  static_assert(INV < 0 || INV == ENF_INV_REL_POS_PERIODIC || INV == ENF_INV_LATITUDE_PERIODIC || INV == ENF_INV_POLAR_PERIODIC ||
                INV == ENF_INV_PONITA, "specialised invariants: two coordinates, no phase");
  const int inv_id = INV >= 0 ? INV : A.inv;
  const int dx_ = INV >= 0 ? 2 : A.dx;
  using Cfg = PairCfg<D, BF16>;
  using SM = PairBwdSmem<D, H, BF16>;
  constexpr int KB = Cfg::KB, NT = Cfg::NT, TT = D / 32;
  constexpr int ST_DD = Cfg::DD::STAGE, ST_GB = Cfg::GB::STAGE, PANEL_GB = Cfg::GB::BYTES;
  using GG = PanelCfg<2 * KB, NT, BF16>;              // one head's d n^ += AGB_h [dgamma; dbeta]
  constexpr int ST_GG = GG::STAGE, PANEL_GG = GG::BYTES;
  constexpr int NW = NWAVES;
  // look-ahead staging: every panel of the z-fold bf16 chain is ONE 32 KB (8 KB at D = 64) stage, so the stage after next
  // can be issued behind each stage's closing barrier; call sites pass `LA ? <stage after next> : <next stage>`
  constexpr bool LA = ZF && BF16 && ENF_K3_LA != 0 && ENF_K3_ANTI == 0 && Cfg::DD::SPP == 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem + SM::RING;
  float* cst = reinterpret_cast<float*>(smem + SM::CONSTS);
  float* c_bq1 = cst, *c_bv1 = cst + D, *c_bf = cst + 2 * D, *c_bm = cst + 3 * D, *c_bgb = cst + 4 * D;
  float* c_acq = c_bgb + 2 * H * D, *c_acv = c_acq + 2 * D;
  char* gcq = smem + SM::GC, *gcv = gcq + SM::GC_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 15, quad = lane >> 4;
  float* zv = reinterpret_cast<float*>(smem + SM::ZVEC) + wave * 2 * H * D;
  const char* blob = A.blob;
  auto G = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };
  // Entry barrier: no wave touches LDS or issues a load before all eight waves of the workgroup are resident (the first four
  // are launched up to ~1000 cycles ahead of the last two, scripts/k3_race/README.md).  Without it the unfolded 64-wide bf16
  // two-head instantiation returned run-to-run different gradients for the latents of waves 4-7 (the first-launched, older
  // wave of each SIMD).  Round 2 established what this is NOT -- not the LDS-DMA ring (read-back, poison and register-staging
  // builds), not a missed barrier (phase self-check), not a stale scalar cache, not a documented MFMA / trans hazard (all
  // measured, scripts/ubench/) -- and found and removed one real defect of the same symptom (inline-asm relu behind
  // compiler-scheduled MFMAs, enf_device.h: relu_f).  The effect itself turned out to be elsewhere: the SLP-packed LayerNorm apply
  // (enf_device.h: ln_apply; scripts/k3_race/README.md, "Resolution") -- this barrier had only moved the timing, and it is compiled out
  // now: 21,000 polluted store-probe calls, 10,000 forward / 3,500 training-backward probe iterations and the GPU suite are clean without it.
  WSTAMP(0);
#if ENF_K3_STATIC_PRIO      // A/B: the second-dispatched half of the workgroup loses every VALU arbitration to its SIMD-mates (T5, static form)
  if (wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);
#endif
#if ENF_K3_ENTRY_BARRIER
  __syncthreads();
#endif

  // this wave's latent: flat (b,z) index; waves past the end keep the barrier cadence only
  // ZF with xcd_remap: workgroup ids are dealt round-robin over the 8 XCDs (id % 8 labels the XCD, MI355X_MICROARCH.md
  // "Workgroup dispatch"), so the nsplit workgroups that sweep the SAME latent's query tiles take consecutive slots of one
  // XCD: they stream that latent's W_zh panels at the same time and the XCD's L2 serves all but the first fetch.  Speed only.
  int bz_ = ZF ? (int)blockIdx.x : blockIdx.x * NW + wave, split_ = blockIdx.y;
  if constexpr (ZF) {
    if (A.xcd_remap) {
      const unsigned k = blockIdx.x >> 3;
      split_ = (int)(k % (unsigned)A.nsplit);
      bz_ = (int)((k / (unsigned)A.nsplit) * 8u + (blockIdx.x & 7u));
    }
  }
  const int bz = bz_;
  const bool active = bz < A.B * A.Z;
  const int bzc = active ? bz : A.B * A.Z - 1;
  const int b = bzc / A.Z;
  const int split = split_;

  for (int i = tid; i < D; i += NTHREADS) { c_bq1[i] = G(A.L.bq1)[i]; c_bv1[i] = G(A.L.bv1)[i]; c_bf[i] = G(A.L.bf)[i]; c_bm[i] = G(A.L.bm)[i]; }
  if constexpr (ZF) { for (int i = tid; i < H * D; i += NTHREADS) c_bgb[i] = G(A.L.p_opbg)[i]; }     // 1 + bgamma_h
  else { for (int i = tid; i < 2 * H * D; i += NTHREADS) c_bgb[i] = G(A.L.bgb)[i]; }
  for (int i = tid; i < 2 * D; i += NTHREADS) { c_acq[i] = G(A.L.acq)[i]; c_acv[i] = G(A.L.acv)[i]; }
  for (int i = tid; i < SM::GC_BYTES / 4; i += NTHREADS) {
    reinterpret_cast<float*>(gcq)[i] = G(A.L.gcq)[i];
    reinterpret_cast<float*>(gcv)[i] = G(A.L.gcv)[i];
  }
  const int ltstride = enf_lt_stride(H, D);
  const float* ltrow = A.lt + (size_t)bzc * ltstride;
  if constexpr (ZF) {          // u | c_zh
    const float* czrow = A.wzb + (size_t)bzc * (H * D);
#pragma unroll
    for (int i = lane * 4; i < H * D; i += 256) {
      *reinterpret_cast<f32x4*>(zv + i) = *reinterpret_cast<const f32x4*>(ltrow + i);
      *reinterpret_cast<f32x4*>(zv + H * D + i) = *reinterpret_cast<const f32x4*>(czrow + i);
    }
  } else {
#pragma unroll
    for (int i = lane * 4; i < 2 * H * D; i += 256)
      *reinterpret_cast<f32x4*>(zv + i) = *reinterpret_cast<const f32x4*>(ltrow + i);
  }
  const f32x4 pz = *reinterpret_cast<const f32x4*>(ltrow + enf_lt_off_pose(H, D));
  const float wcoef = ltrow[enf_lt_off_wcoef(H, D)];
  float cz[H];
#pragma unroll
  for (int h = 0; h < H; ++h) cz[h] = ltrow[enf_lt_off_c(H, D) + h];
  // ball / ball_lat: the latent's rotation matrix and RFF phases (read from the table row: L1/L2 hits), and the
  // per-column partial sums of their gradients
  const bool has_ph = INV < 0 && SM::EXT_OK && enf_inv_has_phase(inv_id);
  const float* ext = ltrow + enf_lt_off_ext(H, D);
  const float* phq = has_ph ? ltrow + enf_lt_off_phq(H, D) : nullptr;
  const float* phv = has_ph ? ltrow + enf_lt_off_phv(H, D) : nullptr;
  float* eacc = reinterpret_cast<float*>(smem + SM::EXTACC) + wave * (11 * 16) + col;
  if (has_ph) {
    if (quad == 0) { for (int k = 0; k < 9; ++k) eacc[k * 16] = 0.f; }
    else if (quad == 1) { eacc[9 * 16] = 0.f; eacc[10 * 16] = 0.f; }
  }

  const unsigned pQ1 = (unsigned)A.L.aq1, pV1 = (unsigned)A.L.av1, pF = (unsigned)A.L.af, pGB = (unsigned)A.L.agb,
                 pM = (unsigned)A.L.am, gQ1 = (unsigned)A.L.gq1, gV1 = (unsigned)A.L.gv1, gF = (unsigned)A.L.gf,
                 gGB = (unsigned)A.L.ggb, gM = (unsigned)A.L.gm;

  Pipe P;
  P.rs = make_blob_rsrc(blob, (unsigned)A.L.total);
  constexpr int PANEL_DD = Cfg::DD::BYTES;
  if constexpr (ZF) P.rs2 = make_blob_rsrc(A.wzt + (size_t)bzc * H * 2 * PANEL_DD, (unsigned)(H * 2 * PANEL_DD));
  else P.rs2 = P.rs;
  const unsigned pWG = (unsigned)A.L.awg;
  first_stage<ST_DD, NW, (ENF_K3_ANTI != 0) && ZF>(P, ring, pQ1, wave, lane);
  if constexpr (LA) stage_issue_p<ST_DD, NW>(P, pV1, ring + STAGE_MAX, lane);       // the second stage is in flight from here on

  // per-lane partial sums over this wave's queries.  dU/dV0: lane (col, quad) holds feature
  // 16 t + col, summed over the queries n = 4 quad + i of every tile (flipped products).
  // ENF_K3_LDSACC: they live in wave-private LDS ([slot][lane], conflict-free) instead of 32 registers that are
  // touched once per tile (and otherwise spilled to scratch for the whole sweep).
#if ENF_K3_LDSACC
  float* lacc = reinterpret_cast<float*>(smem + SM::LACC) + wave * (2 * H * NT * 64) + lane;
  // every lane owns its slots: plain read-modify-write (an LDS float atomic costs ~1000 cycles here)
  // flush NT partial sums of one head at once: all reads, then all adds, then all writes (one LDS round trip)
  auto lacc_flush = [&](int slot0, const float (&part)[NT]) {
    float cur[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cur[t] = lacc[(slot0 + t) * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) lacc[(slot0 + t) * 64] = cur[t] + part[t];
  };
  auto dU_get = [&](int h, int t) { return lacc[(h * NT + t) * 64]; };
  auto dV0_get = [&](int h, int t) { return lacc[((H + h) * NT + t) * 64]; };
#else
  float dU[H][NT], dV0[H][NT];
  auto dU_add = [&](int h, int t, float v) { dU[h][t] += v; };
  auto dV0_add = [&](int h, int t, float v) { dV0[h][t] += v; };
  auto dU_get = [&](int h, int t) { return dU[h][t]; };
  auto dV0_get = [&](int h, int t) { return dV0[h][t]; };
#endif
  float dC[H], dpose[4] = {0.f, 0.f, 0.f, 0.f}, dwc = 0.f;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    dC[h] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#if ENF_K3_LDSACC
      lacc[(h * NT + t) * 64] = 0.f; lacc[((H + h) * NT + t) * 64] = 0.f;
#else
      dU[h][t] = 0.f; dV0[h][t] = 0.f;
#endif
    }
  }

  const int ntiles = (A.N + 15) / 16;
  const int split_tiles = (ntiles - split + A.nsplit - 1) / A.nsplit;  // tiles split, split+nsplit, ..
  // ZF: the 8 waves share the sweep (wave w takes every 8th tile of the split); all run the same number of steps
  const int my_tiles = ZF ? (split_tiles + NW - 1) / NW : split_tiles;
  WSTAMP(1);
#if ENF_K3_PREFETCH
  // this lane's query of tile step `ts` (clamped to a valid row): the per-tile global reads -- coordinates, lse -- are
  // issued one tile ahead (ENF_K3_PREFETCH), under the previous tile's last GEMM stage, instead of at the top of the tile
  // where nothing hides their latency
  auto tile_query = [&](int ts) {
    const int tk_ = ZF ? ts * NW + wave : ts;
    const int n0_ = tk_ < split_tiles ? (split + tk_ * A.nsplit) * 16 : 0;
    return min(n0_ + col, A.N - 1);
  };
  float pf_x[3], pf_lse[H];
  auto prefetch = [&](int ts) {
    const int n_ = tile_query(ts);
    const float* xp = A.x + (size_t)b * A.x_bstride + (size_t)n_ * A.dx;
    pf_x[0] = xp[0]; pf_x[1] = A.dx > 1 ? xp[1] : 0.f; pf_x[2] = A.dx > 2 ? xp[2] : 0.f;
    const size_t qr = (size_t)b * A.N + n_;
#pragma unroll
    for (int h = 0; h < H; ++h) pf_lse[h] = A.lse[qr * H + h];
  };
  if (my_tiles > 0) prefetch(0);
#endif
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tk = ZF ? ti * NW + wave : ti;
    const bool tvalid = tk < split_tiles;
    const int n0 = tvalid ? (split + tk * A.nsplit) * 16 : 0;
    const bool nvalid = tvalid && n0 + col < A.N;
    const int n = min(n0 + col, A.N - 1);
    const size_t qrow = (size_t)b * A.N + n;
#if ENF_K3_PREFETCH
    float t_lse[H];
#pragma unroll
    for (int h = 0; h < H; ++h) t_lse[h] = pf_lse[h];
    const QueryPt q = make_query(pf_x[0], pf_x[1], pf_x[2], inv_id);
#else
    const QueryPt q = load_query(A.x + (size_t)b * A.x_bstride + (size_t)n * dx_, dx_, inv_id);
#endif
    float inv[4], win;
    pair_invariant<BF16>(inv_id, dx_, q, pz, wcoef, A.use_window, inv, win, ext);
    const size_t srow = (size_t)bzc * A.N + n;        // row of the materialised activations (STORE)
    const bool swrite = STORE && nvalid && active;

    BSTAMP(0);
    // ---------------- q-forward: logits -> attention probabilities
    float att[H], dlogit[H];
    Frags<BF16, KB> F;
    unsigned maskq = 0u, maskv = 0u;     // STORE with A.masks: this tile's relu masks (query / value RFFNet layer)
    {
      f32x4 acc[NT];
      rff_embed<D, BF16>(acc, inv, c_acq, lane, quad, phq);
      make_frags<BF16, KB>(F, acc);
      panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_BIAS, LA>(acc, F, P, ring, pQ1, LA ? pF : pV1, true, lane, c_bq1);
      if constexpr (STORE) {
        if (A.masks) {        // relu linearised at the masks' point: h1 = a1 where the bit is set (not max(a1, 0))
          maskq = A.masks[relu_mask_index((b + A.mask_b0) % A.mask_B, A.Z, bzc % A.Z, (A.N + 15) / 16, n0 / 16, 0, lane)];
          maskv = A.masks[relu_mask_index((b + A.mask_b0) % A.mask_B, A.Z, bzc % A.Z, (A.N + 15) / 16, n0 / 16, 1, lane)];
          relu_apply_mask<NT>(acc, maskq);
        }
      }
      const bool masked = STORE && A.masks != nullptr;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float s = tiles_dot<NT>(
            [&](int t) { return masked ? acc[t] : f32x4{relu_f(acc[t][0]), relu_f(acc[t][1]), relu_f(acc[t][2]), relu_f(acc[t][3])}; },
            [&](int t) { return rowvec(zv + h * D, t, quad); });
        const float lg = xquad_sum(s) + cz[h] + win;
        att[h] = __expf(lg - K3_LSE(h));
      }
    }
    BSTAMP(1);
    // ---------------- v-forward to the normalised f
    unsigned relu_mask = 0u;             // bit (4t + i): a2[t][i] > 0
#if ENF_K3_PARK
    Parked<BF16, NT> A3P;                // gelu'(a3) (or a3) waits here for the backward chain in half the registers;
                                         // n^ is recovered from its own fragments F, which the heads keep alive anyway
#else
    f32x4 a3[NT], nh[NT];
#endif
    float mu1, r1;
    {
#if ENF_K3_PARK
      f32x4 a3[NT], nh[NT];
#endif
      f32x4 acc[NT];
      rff_embed<D, BF16>(acc, inv, c_acv, lane, quad, phv);
      make_frags<BF16, KB>(F, acc);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_EV), srow, D, F, quad);
      panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_BIAS, LA>(acc, F, P, ring, pV1, LA ? STAGE_RS2 : pF, true, lane, c_bv1);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (STORE && A.masks) {
            acc[t][i] = (maskv >> (4 * t + i)) & 1u ? acc[t][i] : 0.f;
          } else {
            // relu as an integer maximum (relu_f), the mask bit from its bits: min(bits, 1) is 1 exactly where a2 > 0 -- three
            // instructions per element where compare + select + or + a NaN-quieting fmaxf were five
            if (acc[t][i] > 0.f) relu_mask |= 1u << (4 * t + i);
            acc[t][i] = fmaxf(acc[t][i], 0.f);
          }
        }
      if (STORE && A.masks) relu_mask = maskv;
      make_frags<BF16, KB>(F, acc);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_G1), srow, D, F, quad);
      if constexpr (ZF) panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_BIAS, LA>(a3, F, P, ring, pF, LA ? (ENF_K3_DN_LATE ? gM : STAGE_RS2 | (unsigned)PANEL_DD) : STAGE_RS2, true, lane, c_bf);
      else panel_gemm<KB, NT, BF16, ST_GB, NWAVES, INIT_BIAS>(a3, F, P, ring, pF, pGB, true, lane, c_bf);
#if ENF_K3_A3_FUSED
      // nh = gelu(a3), a3 <- gelu'(a3) from one exp + rcp per element (one tile at a time, as in the heads); the backward
      // needs nothing else of a3, so gelu'(a3) is what gets parked
      K3_SCHED_FENCE();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#if ENF_K3_GELU_PK
        gelu_fg_tile(a3[t], nh[t]);
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float g, d;
          gelu_fg1(a3[t][i], g, d);
          nh[t][i] = g;
          a3[t][i] = d;
        }
#endif
        asm volatile("" : "+v"(nh[t]), "+v"(a3[t]));
        K3_SCHED_FENCE();
      }
#elif ENF_K3_FUSED_GELU
#pragma unroll
      for (int t = 0; t < NT; ++t) nh[t] = a3[t];
      gelu_fg_tiles<NT>(nh, a3);            // nh = gelu(a3); a3 <- gelu'(a3), all the backward needs of it
#else
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nh[t][i] = gelu_f(a3[t][i]);
        K3_SCHED_FENCE();
      }
#endif
      ln_stats<NT>(nh, mu1, r1, A.inv_d);
      ln_apply<NT>(nh, mu1, r1);
      make_frags<BF16, KB>(F, nh);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_NH), srow, D, F, quad);
#if ENF_K3_PARK
      A3P.park(a3);
#endif
    }
    BSTAMP(2);
    f32x4 dnh[NT];                       // d n^ accumulated over heads
    constexpr bool DNL = ZF && ENF_K3_DN_LATE != 0;    // z-fold: d n^ = sum_h W_zh d a5_h is taken AFTER the heads, from their parked d a5
    if constexpr (!DNL) {                              // fragments (16 registers per head instead of the 32 of a running sum)
#pragma unroll
      for (int t = 0; t < NT; ++t) dnh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if constexpr (ZF) {
      Frags<BF16, KB> FAh[DNL ? H : 1];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const unsigned wzf = STAGE_RS2 | (unsigned)(h * 2 * PANEL_DD), wzb = wzf + PANEL_DD;
        // ---- a5 = W_zh^T n + c_zh
        f32x4 a5[NT], v[NT];
#if ENF_K3_ZF_EARLY_DY
        // this head's d ybar row and delta are requested BEFORE the GEMM stage: the L2 round trip runs under its MFMAs and the gelu
        // instead of in front of the dot products that consume it (32 registers that are free while d n^ is taken after the heads)
        f32x4 dy[NT];
        {
          const float* dyrow = A.dybar + qrow * (H * D) + h * D + 4 * quad;
#pragma unroll
          for (int t = 0; t < NT; ++t) dy[t] = *reinterpret_cast<const f32x4*>(dyrow + 16 * t);
        }
        const float delta_h = K3_DELTA(h);
#endif
        panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_BIAS, LA>(a5, F, P, ring, wzf, DNL ? (LA ? pWG + h * PANEL_DD : gM) : (LA ? gM : wzb), true, lane, zv + H * D + h * D);
        BSTAMP(4 + 6 * h);
        float mu2, r2;
#if ENF_K3_ZF_FUSED
        K3_SCHED_FENCE();
        // gelu(a5) and gelu'(a5) share their sigmoid, and nothing but reductions sits between their uses in this branch:
        // one exp + rcp per element, a5 <- gelu'(a5) in place (the same 32 registers it was kept alive in anyway)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#if ENF_K3_GELU_PK
          gelu_fg_tile(a5[t], v[t]);
#else
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float g, d;
            gelu_fg1(a5[t][i], g, d);
            v[t][i] = g;
            a5[t][i] = d;
          }
#endif
          asm volatile("" : "+v"(v[t]), "+v"(a5[t]));
          K3_SCHED_FENCE();
        }
#else
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[t][i] = gelu_f(a5[t][i]);
        }
#endif
        ln_stats<NT>(v, mu2, r2, A.inv_d);
        ln_apply<NT>(v, mu2, r2);
        float s0 = 0.f, sd = 0.f;
#if !ENF_K3_ZF_EARLY_DY
        f32x4 dy[NT];
        const float delta_h = K3_DELTA(h);
        {
          const float* dyrow = A.dybar + qrow * (H * D) + h * D + 4 * quad;
#pragma unroll
          for (int t = 0; t < NT; ++t) dy[t] = *reinterpret_cast<const f32x4*>(dyrow + 16 * t);
        }
#endif
        tiles_sum_dot<NT>(dy, [&](int t) { return v[t]; }, sd, s0);
        const float datt = xquad_sum(s0);
        dlogit[h] = nvalid ? att[h] * (datt - delta_h) : 0.f;
        const float ah = nvalid ? att[h] : 0.f;
        // LayerNorm backward of the weighted cotangent ah * dy: its two means follow from the sums above,
        // mean(ah dy) = ah mean(dy),  mean(ah dy v) = ah datt / D
        const float m1 = ah * xquad_sum(sd) * A.inv_d, m2 = ah * datt * A.inv_d;
        {     // rstd folded into the three per-column scalars: two FMAs and the gelu' product per element
          const float ahr = ah * r2, m1r = m1 * r2, nm2r = -m2 * r2;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            K3_OPAQUE(a5[t]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              dy[t][i] = fmaf(nm2r, v[t][i], fmaf(ahr, dy[t][i], -m1r)) * (ENF_K3_ZF_FUSED ? a5[t][i] : gelu_grad_f(a5[t][i]));   // d a5
          }
        }
        BSTAMP(5 + 6 * h);
        Frags<BF16, KB>& FA = FAh[DNL ? h : 0];
        make_frags<BF16, KB>(FA, dy);
        // ---- d n += W_zh d a5
        if constexpr (!DNL) panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_ACC, LA>(dnh, FA, P, ring, wzb, LA ? pWG + h * PANEL_DD : gM, true, lane);
        // ---- d v0 += sum_n dv (1 + gamma), both as flipped products (rows = queries)
        f32x4 dvf[NT];
        {
          f32x4 none[1];
          const unsigned wzb0 = STAGE_RS2 | (unsigned)PANEL_DD;               // head 0's backward-orientation panel
          // the stage after this head's last one, and the one after that
          const unsigned nxt = DNL ? (h + 1 < H ? wzf + 2 * PANEL_DD : wzb0) : (h + 1 < H ? wzf + 2 * PANEL_DD : gF);
          const unsigned nxt2 = DNL ? (h + 1 < H ? gM : (H > 1 ? wzb0 + 2 * PANEL_DD : gF)) : (h + 1 < H ? wzb + 2 * PANEL_DD : gV1);
          panel_gemm_flip<KB, NT, BF16, ST_DD, NW, false, INIT_ZERO, LA>(
              none, FA, P, ring, gM, LA ? nxt : pWG + h * PANEL_DD, lane, [](int) { return f32x4{0.f, 0.f, 0.f, 0.f}; },
              [&](int mt, const f32x4& af) { dvf[mt] = af; });
          float part[NT];
          panel_gemm_flip<KB, NT, BF16, ST_DD, NW, false, INIT_ACC, LA>(
              none, F, P, ring, pWG + h * PANEL_DD, LA ? nxt2 : nxt, lane,
              [&](int mt) { const float bc = c_bgb[h * D + 16 * mt + col]; return f32x4{bc, bc, bc, bc}; },
              [&](int mt, const f32x4& af) {
#if ENF_K3_DOT_PK
                const f32x2 p2 = __builtin_elementwise_fma(hi2(af), hi2(dvf[mt]), lo2(af) * lo2(dvf[mt]));
                part[mt] = p2[0] + p2[1];
#else
                part[mt] = af[0] * dvf[mt][0] + af[1] * dvf[mt][1] + af[2] * dvf[mt][2] + af[3] * dvf[mt][3];
#endif
              });
#if ENF_K3_LDSACC
          lacc_flush((H + h) * NT, part);
#else
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) dV0_add(h, mt, part[mt]);
#endif
        }
        BSTAMP(6 + 6 * h);
      }
      if constexpr (DNL) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const unsigned wzb = (STAGE_RS2 | (unsigned)(h * 2 * PANEL_DD)) + PANEL_DD;
          const unsigned after = h + 1 < H ? wzb + 2 * PANEL_DD : gF;                                   // next stage
          const unsigned after2 = h + 2 < H ? wzb + 4 * PANEL_DD : (h + 2 == H ? gF : gV1);              // the one after that
          if (h == 0) panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_ZERO, LA>(dnh, FAh[h], P, ring, wzb, LA ? after2 : after, true, lane);
          else panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_ACC, LA>(dnh, FAh[h], P, ring, wzb, LA ? after2 : after, true, lane);
        }
      }
    } else
#pragma unroll
    for (int h = 0; h < H; ++h) {
#if ENF_K3_PARK
      f32x4 v[NT];
      Parked<BF16, NT> OPG;
      {
        f32x4 opgf[NT];
        gb_panel_flip<D, BF16, ST_DD>(v, opgf, F, P, ring, pGB + h * PANEL_GB, pM, c_bgb + 2 * h * D, zv + H * D + h * D,
                                      lane, col, quad);
        OPG.park(opgf);
      }
#else
      f32x4 v[NT], opgf[NT];
      gb_panel_flip<D, BF16, ST_DD>(v, opgf, F, P, ring, pGB + h * PANEL_GB, pM, c_bgb + 2 * h * D, zv + H * D + h * D,
                                    lane, col, quad);
#endif
      BSTAMP(3 + 6 * h);
      f32x4 a5[NT];
      {
        Frags<BF16, KB> FV;
        make_frags<BF16, KB>(FV, v);
        if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_HEAD0 + 4 * h), srow, D, FV, quad);
        panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_BIAS>(a5, FV, P, ring, pM, gM, true, lane, c_bm);
      }
      BSTAMP(4 + 6 * h);
      // d ybar of this head: issued first, the gelu / LayerNorm arithmetic below hides the L2 round trip
      f32x4 dy[NT];
#if ENF_K3_EARLY_DY
      {
        const float* dyrow = A.dybar + qrow * (H * D) + h * D;
#pragma unroll
        for (int t = 0; t < NT; ++t) dy[t] = *reinterpret_cast<const f32x4*>(dyrow + 16 * t + 4 * quad);
      }
#endif
      // mixer LN stats; v <- n~ = (gelu(a5) - mu) * rstd
      float mu2, r2;
#if ENF_K3_UF_FUSED
      // as in the z-fold heads: v = gelu(a5), a5 <- gelu'(a5) from one exp + rcp per element, one tile at a time
      K3_SCHED_FENCE();
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float g, d;
          gelu_fg1(a5[t][i], g, d);
          v[t][i] = g;
          a5[t][i] = d;
        }
        asm volatile("" : "+v"(v[t]), "+v"(a5[t]));
        K3_SCHED_FENCE();
      }
#elif ENF_K3_FUSED_GELU
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = a5[t];
      gelu_fg_tiles<NT>(v, a5);             // v = gelu(a5); a5 <- gelu'(a5)
#else
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[t][i] = gelu_f(a5[t][i]);
        K3_SCHED_FENCE();
      }
#endif
      ln_stats<NT>(v, mu2, r2, A.inv_d);
      ln_apply<NT>(v, mu2, r2);
      // d n~ = att * d ybar ;  d att = d ybar . n~ ;  softmax backward with the forward's lse / delta
#if !ENF_K3_EARLY_DY
      {
        const float* dyrow = A.dybar + qrow * (H * D) + h * D;
#pragma unroll
        for (int t = 0; t < NT; ++t) dy[t] = *reinterpret_cast<const f32x4*>(dyrow + 16 * t + 4 * quad);
      }
#endif
      float s0, sd;
      tiles_sum_dot<NT>(dy, [&](int t) { return v[t]; }, sd, s0);
      const float datt = xquad_sum(s0);
      dlogit[h] = nvalid ? att[h] * (datt - K3_DELTA(h)) : 0.f;
      const float ah = nvalid ? att[h] : 0.f;
      // LayerNorm backward of d n~ = ah * dy (its means follow from the sums above: mean(ah dy) = ah mean(dy),
      // mean(ah dy n~) = ah datt / D), then gelu backward
      const float m1 = ah * xquad_sum(sd) * A.inv_d, m2 = ah * datt * A.inv_d;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // the pre-activation goes through an opaque copy: otherwise hipcc keeps x^2, the exponent and the sigmoid of
        // all 32 elements alive from gelu() above to share them with gelu'() here -- ~100 registers, all spilled
        K3_OPAQUE(a5[t]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          dy[t][i] = r2 * (fmaf(ah, dy[t][i], -m1) - v[t][i] * m2) * ((ENF_K3_FUSED_GELU || ENF_K3_UF_FUSED) ? a5[t][i] : gelu_grad_f(a5[t][i]));   // d a5
        K3_SCHED_FENCE();     // one tile's transcendental chain at a time: interleaving all 32 costs ~100 live registers
      }
      BSTAMP(5 + 6 * h);
      // d v = AM d a5 (transposed, feeds d gamma / d beta) and its flipped twin:
      // d v0[d] += sum_n dv[n][d] (1 + gamma[n][d])
      {
        Frags<BF16, KB> FA;
        make_frags<BF16, KB>(FA, dy);
        if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_HEAD0 + 4 * h + 1), srow, D, FA, quad);
        float part[NT];
        panel_gemm_flip<KB, NT, BF16, ST_GG, NW, true, INIT_ZERO>(
            v, FA, P, ring, gM, gGB + h * PANEL_GG, lane, [](int) { return f32x4{0.f, 0.f, 0.f, 0.f}; },
            [&](int mt, const f32x4& af) {
#if ENF_K3_PARK
              const f32x4 og = OPG.get(mt);
#else
              const f32x4 og = opgf[mt];
#endif
              part[mt] = af[0] * og[0] + af[1] * og[1] + af[2] * og[2] + af[3] * og[3];
            });                                                                                               // v <- d v
#if ENF_K3_LDSACC
        lacc_flush((H + h) * NT, part);
#else
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) dV0_add(h, mt, part[mt]);
#endif
      }
      BSTAMP(6 + 6 * h);
      // FiLM backward: d gamma = d v * v0; d beta = d v.
      // B operand of the [g g b b]-ordered panel: block 2m = d gamma (tiles 2m, 2m+1), block 2m+1 = d beta
      Frags<BF16, 2 * KB> FG;
#pragma unroll
      for (int m = 0; m < KB; ++m) {
        f32x4 dg[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x4 v0 = rowvec(zv + H * D + h * D, 2 * m + e, quad);
#pragma unroll
          for (int i = 0; i < 4; ++i) dg[e][i] = v[2 * m + e][i] * v0[i];
        }
        if constexpr (BF16) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            FG.f[2 * m][j] = (__bf16)dg[j >> 2][j & 3];
            FG.f[2 * m + 1][j] = (__bf16)v[2 * m + (j >> 2)][j & 3];
          }
        } else {
          FG.f[4 * m] = dg[0]; FG.f[4 * m + 1] = dg[1];
          FG.f[4 * m + 2] = v[2 * m]; FG.f[4 * m + 3] = v[2 * m + 1];
        }
      }
      BSTAMP(7 + 6 * h);
      if (swrite) {
        Frags<BF16, KB> Fg, Fb;
#pragma unroll
        for (int m = 0; m < KB; ++m) {
          if constexpr (BF16) { Fg.f[m] = FG.f[2 * m]; Fb.f[m] = FG.f[2 * m + 1]; }
          else { Fg.f[2 * m] = FG.f[4 * m]; Fg.f[2 * m + 1] = FG.f[4 * m + 1]; Fb.f[2 * m] = FG.f[4 * m + 2]; Fb.f[2 * m + 1] = FG.f[4 * m + 3]; }
        }
        store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_HEAD0 + 4 * h + 2), srow, D, Fg, quad);
        store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_HEAD0 + 4 * h + 3), srow, D, Fb, quad);
      }
      if (h + 1 < H) panel_gemm<2 * KB, NT, BF16, ST_GB>(dnh, FG, P, ring, gGB + h * PANEL_GG, pGB + (h + 1) * PANEL_GB, true, lane);
      else panel_gemm<2 * KB, NT, BF16, ST_DD>(dnh, FG, P, ring, gGB + h * PANEL_GG, gF, true, lane);
    }

    BSTAMP(15);
    // ---------------- LN / gelu backward -> d a3 -> AF -> relu -> W1v -> d E_v -> d t_v -> d inv
    float dinv[4] = {0.f, 0.f, 0.f, 0.f};
    float dlat[2] = {0.f, 0.f};           // ball / ball_lat: gradient rows 4, 5 of the gc panels (quad-1 lanes)
    {
      float s1, s2;
#if ENF_K3_PARK
      tiles_sum_dot<NT>(dnh, [&](int t) { return Parked<BF16, NT>{F}.get(t); }, s1, s2);
#else
      tiles_sum_dot<NT>(dnh, [&](int t) { return nh[t]; }, s1, s2);
#endif
      const float m1 = xquad_sum(s1) * A.inv_d, m2 = xquad_sum(s2) * A.inv_d;
      const float m1r = m1 * r1, nm2r = -m2 * r1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#if ENF_K3_PARK
        const f32x4 nht = Parked<BF16, NT>{F}.get(t);
        f32x4 a3t = A3P.get(t);
        K3_OPAQUE(a3t);
#else
        const f32x4 nht = nh[t], a3t = a3[t];
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
          dnh[t][i] = fmaf(nm2r, nht[i], fmaf(r1, dnh[t][i], -m1r)) * ((ENF_K3_FUSED_GELU || ENF_K3_A3_FUSED) ? a3t[i] : gelu_grad_f(a3t[i]));   // d a3
        K3_SCHED_FENCE();
      }
      make_frags<BF16, KB>(F, dnh);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_DA3), srow, D, F, quad);
      f32x4 acc[NT];
      panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_ZERO, LA>(acc, F, P, ring, gF, LA ? pQ1 : gV1, true, lane);       // d g1
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)       // d a2
          acc[t][i] = ((relu_mask >> (4 * t + i)) & 1u) ? acc[t][i] : 0.f;
      make_frags<BF16, KB>(F, acc);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_DA2), srow, D, F, quad);
      panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_ZERO, LA>(acc, F, P, ring, gV1, LA ? gQ1 : pQ1, true, lane);      // d E_v
      f32x4 Ev[NT];
      rff_embed<D, BF16>(Ev, inv, c_acv, lane, quad, phv);                                                           // recomputed
      f32x4 dT[TT];
      rff_embed_bwd<D>(dT, acc, Ev);
      Frags<BF16, D / 64> FT;
      make_frags<BF16, D / 64>(FT, dT);
      f32x4 di[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      gemm_stage<BF16, D / 64, 1>(di, FT, gcv, lane);
      if (quad == 0) { dinv[0] += di[0][0]; dinv[1] += di[0][1]; dinv[2] += di[0][2]; dinv[3] += di[0][3]; }
      else if (quad == 1) { dlat[0] += di[0][0]; dlat[1] += di[0][1]; }
    }
    BSTAMP(16);
    // ---------------- q-branch: recompute a1 (transposed, for the relu mask of d h1) and its flipped
    // twin h1f (rows = queries) for d u[f] += sum_n dlogit[n] h1f[n][f]
    {
      f32x4 E[NT];
      rff_embed<D, BF16>(E, inv, c_acq, lane, quad, phq);
      make_frags<BF16, KB>(F, E);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_EQ), srow, D, F, quad);
      f32x4 acc[NT];
      float dl[H][4];                     // dlogit of the 4 queries this lane's flipped rows hold
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) dl[h][i] = __shfl(dlogit[h], (quad << 4) | (4 * quad + i), 64);
      const bool more = ti + 1 < my_tiles;
      float upart[H][NT];
      // flipped tiles hold feature 16 mt + col of queries 4 quad + i: their mask bits sit in the words of lanes
      // (col >> 2) * 16 + 4 quad + i, bit 4 mt + (col & 3)
      unsigned mflip[4] = {0u, 0u, 0u, 0u};
      const bool masked = STORE && A.masks != nullptr;
      if (masked) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mflip[i] = (unsigned)__shfl((int)maskq, ((col >> 2) << 4) | (4 * quad + i), 64) >> (col & 3);
      }
      panel_gemm_flip<KB, NT, BF16, ST_DD, NW, true, INIT_BIAS, LA>(
          acc, F, P, ring, pQ1, LA ? (more ? pQ1 : NO_STAGE) : gQ1, lane,
          [&](int mt) { const float bc = c_bq1[16 * mt + col]; return f32x4{bc, bc, bc, bc}; },
          [&](int mt, const f32x4& af) {
            float r0, r1, r2, r3;
            if (masked) {
              r0 = (mflip[0] >> (4 * mt)) & 1u ? af[0] : 0.f; r1 = (mflip[1] >> (4 * mt)) & 1u ? af[1] : 0.f;
              r2 = (mflip[2] >> (4 * mt)) & 1u ? af[2] : 0.f; r3 = (mflip[3] >> (4 * mt)) & 1u ? af[3] : 0.f;
            } else { r0 = relu_f(af[0]); r1 = relu_f(af[1]); r2 = relu_f(af[2]); r3 = relu_f(af[3]); }
#pragma unroll
            for (int h = 0; h < H; ++h) {
#if ENF_K3_DOT_PK
              const f32x2 p2 = __builtin_elementwise_fma(f32x2{r2, r3}, f32x2{dl[h][2], dl[h][3]}, f32x2{r0, r1} * f32x2{dl[h][0], dl[h][1]});
              upart[h][mt] = p2[0] + p2[1];
#else
              upart[h][mt] = dl[h][0] * r0 + dl[h][1] * r1 + dl[h][2] * r2 + dl[h][3] * r3;
#endif
            }
          }, c_bq1);                                                                                             // a1
#pragma unroll
      for (int h = 0; h < H; ++h) {
#if ENF_K3_LDSACC
        lacc_flush(h * NT, upart[h]);
#else
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) dU_add(h, mt, upart[h][mt]);
#endif
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const f32x4 u = rowvec(zv + h * D, t, quad);
#pragma unroll
          for (int i = 0; i < 4; ++i) dh[i] = fmaf(dlogit[h], u[i], dh[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[t][i] = (masked ? ((maskq >> (4 * t + i)) & 1u) != 0u : acc[t][i] > 0.f) ? dh[i] : 0.f;             // d a1
      }
      Frags<BF16, KB> FA;
      make_frags<BF16, KB>(FA, acc);
      if (swrite) store_frags<BF16, KB>(k3_store_ptr(A, ENF_S_DA1), srow, D, FA, quad);
#if ENF_K3_PREFETCH
      if (more) prefetch(ti + 1);          // the next tile's coordinates / lse land under this stage
#endif
      panel_gemm<KB, NT, BF16, ST_DD, NWAVES, INIT_ZERO, LA>(acc, FA, P, ring, gQ1, more ? (LA ? pV1 : pQ1) : NO_STAGE, true, lane);   // d E_q
      f32x4 dT[TT];
      rff_embed_bwd<D>(dT, acc, E);
      Frags<BF16, D / 64> FT;
      make_frags<BF16, D / 64>(FT, dT);
      f32x4 di[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      gemm_stage<BF16, D / 64, 1>(di, FT, gcq, lane);
      if (quad == 0) { dinv[0] += di[0][0]; dinv[1] += di[0][1]; dinv[2] += di[0][2]; dinv[3] += di[0][3]; }
      else if (quad == 1) { dlat[0] += di[0][0]; dlat[1] += di[0][1]; }
    }
    BSTAMP(17);
    // ---------------- per-latent scalars (each column is counted once: quad 0)
    if (quad == 0) {
      float dwin = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) { dC[h] += dlogit[h]; dwin += dlogit[h]; }
      float dxv[3];
      float* dxp = A.dxq ? dxv : nullptr;
      if (has_ph) {
        float dR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        pair_invariant_bwd<BF16>(inv_id, dx_, q, pz, wcoef, A.use_window, inv, win, dinv, dwin, dpose, dwc, dR, dxp, ext);
        if (inv_id == ENF_INV_BALL) {
#pragma unroll
          for (int k = 0; k < 9; ++k) eacc[k * 16] += dR[k];
        }
      } else
      pair_invariant_bwd<BF16>(inv_id, dx_, q, pz, wcoef, A.use_window, inv, win, dinv, dwin, dpose, dwc, nullptr, dxp);
      if (A.dxq && nvalid && active) {          // every latent's wave adds its share to the query's gradient
        float* o = A.dxq + ((size_t)b * A.N + n) * dx_;
        atomicAdd(o, dxv[0]);
        if (dx_ > 1) atomicAdd(o + 1, dxv[1]);
        if (dx_ > 2) atomicAdd(o + 2, dxv[2]);
      }
    } else if (has_ph && quad == 1) { eacc[9 * 16] += dlat[0]; eacc[10 * 16] += dlat[1]; }
  }

#ifdef ENF_TEST_HOOKS
  if constexpr (ENF_K3_LDSACC != 0 && 2 * H * (D / 16) <= 16) {
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < ENF_HOOK_WGS) {
      float* o = enf_hook_wave_sums + (size_t)(wg * NWAVES + wave) * ENF_HOOK_ROW;
      for (int k = 0; k < 2 * H * NT; ++k) o[k * 64 + lane] = lacc[k * 64];
      float q0[H + 5];
      for (int h = 0; h < H; ++h) q0[h] = dC[h];
      q0[H] = dpose[0]; q0[H + 1] = dpose[1]; q0[H + 2] = dpose[2]; q0[H + 3] = dpose[3]; q0[H + 4] = dwc;
      for (int i = 0; i < H + 5; ++i) {
        float a = quad == 0 ? q0[i] : 0.f;
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) o[16 * 64 + i] = a;
      }
      if (lane == 0) { o[16 * 64 + 8] = (float)bz; o[16 * 64 + 9] = (float)bzc; o[16 * 64 + 10] = (float)my_tiles; }
    }
  }
#endif
  pipe_finish(P);          // (antiphase staging only: the early waves take the last stage's barrier here)
  WSTAMP(2);
  // ---- fold the partial sums and add this wave's share into the latent-table gradient
  if (!active) return;   // (unfolded: a wave without a latent; no barrier follows on that path.  z-fold: all eight waves share the latent)
  float* drow = A.dlt + (size_t)bz * ltstride;
#if ENF_K3_LDSACC
  if constexpr (ZF) {
    // the eight waves of a z-fold workgroup share one latent: their partial sums are added through LDS first -- wave w folds
    // slots w, w + 8, ... over the waves (fixed order) and the quads -- so a gradient element gets ONE atomic per workgroup
    // (one per launch with nsplit = 1: then the sum is order-independent) instead of eight on the same address
    __syncthreads();
    const float* lall = reinterpret_cast<const float*>(smem + SM::LACC) + lane;      // [wave][slot][lane]
    constexpr int NS = 2 * H * NT;
    for (int sl = wave; sl < NS; sl += NW) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += lall[(w * NS + sl) * 64];
      v = xquad_sum(v);
      if (quad == 0)
        atomicAdd(drow + (sl < H * NT ? enf_lt_off_u(H, D) + 16 * sl : enf_lt_off_v0(H, D) + 16 * (sl - H * NT)) + col, v);
    }
  } else
#endif
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float a = xquad_sum(dU_get(h, t)), c = xquad_sum(dV0_get(h, t));
      if (quad == 0) {
        atomicAdd(drow + enf_lt_off_u(H, D) + h * D + 16 * t + col, a);
        atomicAdd(drow + enf_lt_off_v0(H, D) + h * D + 16 * t + col, c);
      }
    }
  float sc[H + 5];
#pragma unroll
  for (int h = 0; h < H; ++h) sc[h] = dC[h];
  sc[H] = dpose[0]; sc[H + 1] = dpose[1]; sc[H + 2] = dpose[2]; sc[H + 3] = dpose[3]; sc[H + 4] = dwc;
#pragma unroll
  for (int i = 0; i < H + 5; ++i) {
    float a = quad == 0 ? sc[i] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    sc[i] = a;
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < H; ++h) atomicAdd(drow + enf_lt_off_c(H, D) + h, sc[h]);
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(drow + enf_lt_off_pose(H, D) + i, sc[H + i]);
    atomicAdd(drow + enf_lt_off_wcoef(H, D), sc[H + 4]);
  }
  if (has_ph) {            // d R (9) | d(latent-only invariants) (2) -> the gradient row's ext field
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      float a = quad == (k < 9 ? 0 : 1) ? eacc[k * 16] : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0) atomicAdd(drow + enf_lt_off_ext(H, D) + k, a);
    }
  }
  WSTAMP(3);
}

template <int D, int H, bool BF16, bool STORE, bool ZF, int INV = -1>
static int launch_pair_bwd(const PairBwdArgs& A, hipStream_t st) {
  using SM = PairBwdSmem<D, H, BF16>;
  auto kern = enf_pair_bwd_kernel<D, H, BF16, STORE, ZF, INV>;
  static EnfAttrBits attr_done{0};          // one per instantiation, one bit per device
  if (!enf_lds_attr(reinterpret_cast<const void*>(kern), SM::TOTAL, attr_done)) return ENF_ELAUNCH;
  dim3 grid(ZF ? A.B * A.Z : (A.B * A.Z + NWAVES - 1) / NWAVES, A.nsplit);
  if (ZF && A.xcd_remap) grid = dim3(A.B * A.Z * A.nsplit, 1);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), SM::TOTAL, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

// relu masks: per call (EnfDims.masks / mask_mode / mask_B, from the descriptor); read by the STORE instantiation only

extern "C" int enf_launch_pair_bwd(const EnfDims& m, const EnfLayout& L, const char* blob, const float* x, long long x_bstride,
                                   const float* lt, const float* lse, const float* dybar, const float* delta, float* dlt,
                                   void* const* store, const char* wzt, const float* wzb, float* dxq, hipStream_t st) {
  PairBwdArgs A;
  A.dxq = dxq;
  A.masks = store && m.mask_mode == ENF_MASK_READ ? m.masks : nullptr; A.mask_B = m.mask_B; A.mask_b0 = m.mask_b0;
  const bool zf = !store && wzt && wzb && (size_t)m.H * 2 * enf_panel_bytes(m.D, m.D, m.bf16) < 0x7fffffffu;
  A.wzt = wzt; A.wzb = wzb; A.inv_d = 1.0f / (float)m.Dt;
  A.x = x; A.x_bstride = x_bstride; A.lt = lt; A.blob = blob; A.L = L; A.lse = lse; A.dybar = dybar; A.delta = delta;
  A.dlt = dlt; A.B = m.B; A.N = m.N; A.Z = m.Z; A.dx = m.dx; A.inv = m.inv; A.use_window = m.use_window;
  // one workgroup per CU is resident (LDS): split the query tiles over grid.y until all 256 CUs have one
  const int wgs = zf ? m.B * m.Z : (m.B * m.Z + NWAVES - 1) / NWAVES, ntiles = (m.N + 15) / 16;
  int ns = 1;
  while (wgs * ns < 256 && ns * 2 <= ntiles) ns *= 2;
#ifdef ENF_K3_ZF_NSPLIT      // A/B builds: force the z-fold split
  if (zf) { ns = ENF_K3_ZF_NSPLIT; while (ns > 1 && ns > ntiles) ns /= 2; }
#else
  // z-fold, chip already full: split each latent's sweep over ENF_K3_ZF_SPLIT workgroups placed on one XCD, so that the XCD's
  // resident workgroups stream 32 / split latents' panels (128 KB each at D = 128, H = 2) instead of 32 -- they then stay in
  // the 4 MB L2 across the sweep steps instead of coming from HBM every step
  if (zf && ns == 1 && wgs % 8 == 0 && wgs >= 256) { ns = ENF_K3_ZF_SPLIT; while (ns > 1 && ns * 8 > ntiles) ns /= 2; }
#endif
  A.nsplit = ns;
  A.xcd_remap = zf && ns > 1 && wgs % 8 == 0 && ENF_K3_XCD_REMAP;
  if (store)
    for (int i = 0; i < ENF_NUM_STORE(m.H); ++i) A.store[i] = store[i];
#define ENF_CASE(DD, HH)                                                                   \
  if (m.D == DD && m.H == HH) {                                                            \
    if (store) return m.bf16 ? launch_pair_bwd<DD, HH, true, true, false>(A, st) : launch_pair_bwd<DD, HH, false, true, false>(A, st); \
    if (zf) return m.bf16 ? launch_pair_bwd<DD, HH, true, false, true>(A, st) : launch_pair_bwd<DD, HH, false, false, true>(A, st);    \
    return m.bf16 ? launch_pair_bwd<DD, HH, true, false, false>(A, st) : launch_pair_bwd<DD, HH, false, false, false>(A, st);         \
  }
#if ENF_K3_INV_SPECIALISED
  // the z-fold bf16 kernels of the shipped configs with the invariant as a compile-time constant (configs 2, 4, 5:
  // rel_pos_periodic; config 3: latitude_periodic and the SO(3) polar_periodic; config 1: ponita at num_hidden 64)
  if (zf && m.bf16 && m.dx == 2) {
    if (m.D == 128 && m.H == 2) {
      if (m.inv == ENF_INV_REL_POS_PERIODIC) return launch_pair_bwd<128, 2, true, false, true, ENF_INV_REL_POS_PERIODIC>(A, st);
      if (m.inv == ENF_INV_LATITUDE_PERIODIC) return launch_pair_bwd<128, 2, true, false, true, ENF_INV_LATITUDE_PERIODIC>(A, st);
      if (m.inv == ENF_INV_POLAR_PERIODIC) return launch_pair_bwd<128, 2, true, false, true, ENF_INV_POLAR_PERIODIC>(A, st);
    }
    if (m.D == 64 && m.H == 2 && m.inv == ENF_INV_PONITA) return launch_pair_bwd<64, 2, true, false, true, ENF_INV_PONITA>(A, st);
  }
#if ENF_K3_STORE_SPEC
  // the training path's kernel (activation store) of the headline configs likewise
  if (store && m.bf16 && m.dx == 2 && m.D == 128 && m.H == 2 && m.inv == ENF_INV_REL_POS_PERIODIC)
    return launch_pair_bwd<128, 2, true, true, false, ENF_INV_REL_POS_PERIODIC>(A, st);
#endif
#endif
  ENF_CASE(128, 2)
  ENF_CASE(64, 2)
  ENF_CASE(128, 1)
  ENF_CASE(64, 1)
  ENF_CASE(64, 4)
#undef ENF_CASE
  return ENF_EUNSUPPORTED;
}
