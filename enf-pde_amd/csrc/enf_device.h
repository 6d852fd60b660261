// enf_device.h -- device building blocks shared by the pair / tail kernels (gfx950 only).
//
// Activation layout ("acc layout"): a 32-feature x 32-column tile lives in one f32x16 per
// lane, exactly as v_mfma_f32_32x32x* writes its C/D operand:
//     column = lane & 31,  feature row = RHO(reg, lane >> 5) = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Columns are pairs (query n, latent z) or queries; features are network channels.  Every layer
// is computed TRANSPOSED, Y^T = W^T X^T, so the previous layer's accumulator is directly the
// next MFMA's B operand (cdna_hip_programming.md section 3, "An accumulator tile as the next
// MFMA's operand"): no LDS round trip and no cross-lane traffic between layers.  The weight
// panels are pre-packed in A-operand fragment order with the matching k permutation (enf_pack.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#define DEV __device__ __forceinline__

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

DEV constexpr int RHO(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------ B-operand fragments
template <bool BF16, int KB>
struct Frags {
  using T = typename std::conditional<BF16, bf16x8, f32x16>::type;
  T f[BF16 ? 2 * KB : KB];
};

// X: KB blocks in acc layout -> fragments of the next layer's B operand.
// bf16: element j of k-step s of block blk is X[blk][8s+j]  (feature 32blk + 16s + 8(j>>2) + 4h + (j&3))
// fp32: MFMA #(blk, r) takes X[blk][r] as is (k = lane>>5 <-> feature 32blk + RHO(r, lane>>5)).
template <bool BF16, int KB>
DEV void make_frags(Frags<BF16, KB>& F, const f32x16 (&X)[KB]) {
#pragma unroll
  for (int blk = 0; blk < KB; ++blk) {
    if constexpr (BF16) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) F.f[2 * blk + s][j] = (__bf16)X[blk][8 * s + j];
    } else {
      F.f[blk] = X[blk];
    }
  }
}

// acc[m] += W^T[32m.., :] X  for MBS out-blocks whose fragments start at `lds`
// (panel order: [m][blk][s or r4][lane] x 16 bytes; see enf_pack.hip).
template <bool BF16, int KB, int MBS>
DEV void gemm_stage(f32x16* acc, const Frags<BF16, KB>& F, const char* lds, int lane) {
#pragma unroll
  for (int m = 0; m < MBS; ++m) {
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
      if constexpr (BF16) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + ((((m * KB + blk) * 2 + s) * 64 + lane) << 4));
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, F.f[2 * blk + s], acc[m], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(lds + ((((m * KB + blk) * 4 + r4) * 64 + lane) << 4));
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], F.f[blk][4 * r4 + i], acc[m], 0, 0, 0);
        }
      }
    }
    // keep the A-fragment ds_reads of later out-blocks from being hoisted above this block's MFMAs
    // (the scheduler otherwise clusters them all up front and spills)
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool BF16> constexpr int frag_bytes() { return BF16 ? 2048 : 4096; }

// per-row constant vector (bias, u, v0 ..) -> acc layout; `vec` is fp32 in LDS or global,
// the address depends on the lane only through its half, so the read is a broadcast.
DEV void load_rowvec(f32x16& acc, const float* vec, int blk, int half) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(vec + 32 * blk + 8 * g + 4 * half);
    acc[4 * g + 0] = v[0]; acc[4 * g + 1] = v[1]; acc[4 * g + 2] = v[2]; acc[4 * g + 3] = v[3];
  }
}

// sum over both lane halves (the two halves of a column hold disjoint feature rows)
DEV float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// ------------------------------------------------------------------ math
// gelu, tanh approximation (jax.nn.gelu default): 0.5x(1+tanh(c(x+0.044715x^3))) = x*sigmoid(2c(..))
DEV float gelu_f(float x) {
  const float c2 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2*sqrt(2/pi)*log2(e)
  const float u = x * (c2 + (c2 * 0.044715f) * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
}
// d/dx of the above: s + x s (1-s) 2c(1+3*0.044715x^2),  s = sigmoid(2c(x+0.044715x^3))
DEV float gelu_grad_f(float x) {
  const float c = 0.7978845608028654f;
  const float c2 = -2.0f * c * 1.4426950408889634f;
  const float x2 = x * x;
  const float u = x * (c2 + (c2 * 0.044715f) * x2);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
  return s + x * s * (1.0f - s) * (2.0f * c) * (1.0f + 3.0f * 0.044715f * x2);
}

// sin / cos of 2*pi*t (t in revolutions).  bf16 mode: the hardware v_sin/v_cos take revolutions.
template <bool FAST> DEV float sin_rev(float t) {
  if constexpr (FAST) return __builtin_amdgcn_sinf(t);
  else return sinpif(2.0f * t);
}
template <bool FAST> DEV float cos_rev(float t) {
  if constexpr (FAST) return __builtin_amdgcn_cosf(t);
  else return cospif(2.0f * t);
}

// ------------------------------------------------------------------ weight staging
// All 256 threads copy one stage (<= STAGE_MAX bytes, multiple of 4 KB) global -> registers
// (issue, before the compute that hides the latency) -> LDS (commit, after the compute).
constexpr int STAGE_MAX = 32768;
constexpr unsigned NO_STAGE = 0xFFFFFFFFu;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Stage sources are byte offsets into the packed blob.  A stage moves global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds): each wave-instruction carries 1 KB (lane-linear, which is exactly
// the fragment order the panels are packed in), no VGPR is touched, descriptor and offsets are
// scalar (cdna_hip_programming.md section 5, T8).  The DMA is issued BEFORE the MFMAs that hide its
// latency and retired by stage_wait() + the stage barrier.
DEV __amdgpu_buffer_rsrc_t make_blob_rsrc(const char* blob, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blob), 0, bytes, 0x00020000);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BYTES>
DEV void stage_issue(__amdgpu_buffer_rsrc_t rs, unsigned src_off, char* dst, int wave, int lane) {
  static_assert(BYTES % 4096 == 0 && BYTES <= STAGE_MAX, "stage size");
#pragma unroll
  for (int i = 0; i < BYTES / 4096; ++i) {
    const int piece = (i * 4 + wave) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + piece), 16, lane * 16, src_off + piece, 0, 0);
  }
}
DEV void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Panel = (MBOUT out-blocks) x (KBIN in-blocks) of A-operand fragments, streamed through a
// 2-slot LDS ring (slot stride STAGE_MAX) in stages of MBS out-blocks.
struct Pipe { __amdgpu_buffer_rsrc_t rs; int cur; int wave; };

template <int KBIN, int MBOUT, bool BF16> struct PanelCfg {
  static constexpr int FRAG = BF16 ? 2048 : 4096;
  static constexpr int MBLK_BYTES = KBIN * FRAG;
  static constexpr int MBS = (STAGE_MAX / MBLK_BYTES) < MBOUT ? (STAGE_MAX / MBLK_BYTES) : MBOUT;
  static_assert(MBS >= 1 && MBOUT % MBS == 0, "panel staging");
  static constexpr int SPP = MBOUT / MBS;           // stages per panel
  static constexpr int STAGE = MBS * MBLK_BYTES;    // bytes per stage
  static constexpr int BYTES = MBOUT * MBLK_BYTES;
};

// acc[MBOUT] += panel . F.  Precondition: the panel's first stage is resident in ring[cur] and
// visible (a barrier has passed).  While stage s is multiplied, stage s+1 (or the first stage of
// `next`, NEXT_BYTES long; NO_STAGE = nothing follows) streams into the other ring slot; one
// wait + one barrier per stage publish it.  `panel` / `next` are blob byte offsets (wave-uniform).
// `active` (wave-uniform) lets a wave without work keep the staging / barrier cadence.
template <int KBIN, int MBOUT, bool BF16, int NEXT_BYTES>
DEV void panel_gemm(f32x16 (&acc)[MBOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel,
                    unsigned next, bool active, int tid, int lane) {
  using C = PanelCfg<KBIN, MBOUT, BF16>;
#pragma unroll
  for (int sp = 0; sp < C::SPP; ++sp) {
    if (sp + 1 < C::SPP) stage_issue<C::STAGE>(P.rs, panel + (sp + 1) * C::STAGE, ring + (P.cur ^ 1) * STAGE_MAX, P.wave, lane);
    else if (next != NO_STAGE) stage_issue<NEXT_BYTES>(P.rs, next, ring + (P.cur ^ 1) * STAGE_MAX, P.wave, lane);
    if (active) gemm_stage<BF16, KBIN, C::MBS>(&acc[sp * C::MBS], F, ring + P.cur * STAGE_MAX, lane);
    stage_wait();
    __syncthreads();
    P.cur ^= 1;
  }
}

// LayerNorm statistics over the KB*32 features of this lane's column (biased variance, eps 1e-6)
template <int KB> DEV void ln_stats(const f32x16 (&X)[KB], float& mu, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += X[b][r];
  mu = xhalf_sum(s) * (1.0f / (32 * KB));
  float q = 0.f;
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float t = X[b][r] - mu; q = fmaf(t, t, q); }
  rstd = rsqrtf(xhalf_sum(q) * (1.0f / (32 * KB)) + 1e-6f);
}

// ------------------------------------------------------------------ invariants + window
// xq: query coordinate (dx <= 3) of this lane's column; pz: latent pose row from the latent
// table: periodic/rel/abs/norm -> (p0,p1,p2,-); ponita -> (px,py,cos t,sin t);
// sphere -> (phi, theta, sin theta, cos theta).  sx/cx = sin/cos(theta_x) (sphere only).
struct QueryPt { float x0, x1, x2, sx, cx; };

template <bool FAST>
DEV void pair_invariant(int inv_id, int dx, const QueryPt& q, const f32x4& pz, float wcoef, int use_window,
                        float (&inv)[4], float& win) {
  inv[0] = inv[1] = inv[2] = inv[3] = 0.f;
  win = 0.f;
  switch (inv_id) {
    case ENF_INV_REL_POS_PERIODIC: {          // rel_pos_periodic.py:47-60; window _base_invariant.py:35-43
      const float d0 = pz[0] - q.x0, d1 = pz[1] - q.x1;
      inv[0] = cos_rev<FAST>(0.5f * d0); inv[1] = cos_rev<FAST>(0.5f * d1);
      inv[2] = sin_rev<FAST>(0.5f * d0); inv[3] = sin_rev<FAST>(0.5f * d1);
      if (use_window) win = wcoef * (inv[0] * inv[0] + inv[1] * inv[1]);
    } break;
    case ENF_INV_LATITUDE_PERIODIC:            // spherical_longitude.py:68-85; window :34-55
    case ENF_INV_POLAR_PERIODIC: {             // polar_periodic.py:52-68;      window :35-38
      const float dphi = (q.x0 - pz[0]) * 0.15915494309189535f;   // revolutions
      const float cd = cos_rev<FAST>(dphi), sd = sin_rev<FAST>(dphi);
      const float dot = q.sx * pz[2] * cd + q.cx * pz[3];
      if (inv_id == ENF_INV_LATITUDE_PERIODIC) { inv[0] = q.x1; inv[1] = pz[1]; inv[2] = cd; inv[3] = sd; }
      else inv[0] = dot;
      if (use_window) {
        const float dc = fminf(fmaxf(dot, -1.f + 1e-6f), 1.f - 1e-6f);
        const float ang = acosf(dc);
        win = __expf(-ang * ang * wcoef);
      }
    } break;
    case ENF_INV_PONITA: {                     // ponita.py:30-44; window _base_invariant.py:25-33
      const float r0 = q.x0 - pz[0], r1 = q.x1 - pz[1];
      inv[0] = r0 * pz[2] + r1 * pz[3];
      inv[1] = -r0 * pz[3] + r1 * pz[2];
      if (use_window) win = -wcoef * (r0 * r0 + r1 * r1);
    } break;
    default: {                                 // abs_pos.py:42, rel_pos.py:41, norm_rel_pos.py:34
      const float r0 = q.x0 - pz[0], r1 = dx > 1 ? q.x1 - pz[1] : 0.f, r2 = dx > 2 ? q.x2 - pz[2] : 0.f;
      const float d2 = r0 * r0 + r1 * r1 + r2 * r2;
      if (inv_id == ENF_INV_ABS_POS) { inv[0] = q.x0; inv[1] = q.x1; inv[2] = q.x2; }
      else if (inv_id == ENF_INV_REL_POS) { inv[0] = r0; inv[1] = r1; inv[2] = r2; }
      else inv[0] = sqrtf(d2);
      if (use_window) win = -wcoef * d2;
    } break;
  }
}
