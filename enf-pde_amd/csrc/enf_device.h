// enf_device.h -- device building blocks shared by the pair / tail kernels (gfx950 only).
//
// Activation layout ("acc layout"): a TILE = 16 features x 16 columns lives in one f32x4 per
// lane, exactly as v_mfma_f32_16x16x* writes its C/D operand:
//     column = lane & 15,   feature row = 4*(lane >> 4) + reg          (quad q = lane >> 4)
// A D-wide activation is NT = D/16 tiles (f32x4 X[NT]); two consecutive tiles form a 32-feature
// BLOCK, the K extent of one bf16 MFMA.  Columns are pairs (query n, latent z) or queries.
// Every layer is computed TRANSPOSED, Y^T = W^T X^T, so the previous layer's accumulator is
// directly the next MFMA's B operand (cdna_hip_programming.md section 3, "An accumulator tile as
// the next MFMA's operand"): no LDS round trip and no cross-lane traffic between layers.  The
// weight panels are pre-packed in A-operand fragment order with the matching k permutation
// (enf_pack.hip).  One wave owns 16 columns; a workgroup is 8 waves = 2 per SIMD, which keeps
// every wave under 256 registers, doubles the VALU issue rate a lone wave gets and lets one
// wave's VALU epilogue run under the other's MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#define DEV __device__ __forceinline__

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWAVES = 8;            // waves per workgroup
constexpr int NTHREADS = 64 * NWAVES;

// ------------------------------------------------------------------ B-operand fragments
// bf16: one bf16x8 per 32-feature block; element j (k = 8q + j) is feature 16(j>>2) + 4q + (j&3),
//       i.e. tile (j>>2) register (j&3) of this lane.
// fp32: v_mfma_f32_16x16x4_f32 #(tile, i) takes X[tile][i] as is (k = q <-> feature 16 tile + 4q + i).
template <bool BF16, int KB>
struct Frags {
  using T = typename std::conditional<BF16, bf16x8, f32x4>::type;
  T f[BF16 ? KB : 2 * KB];
};

// two floats -> one word of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32.  Written element by element
// (`f[j] = (__bf16)x`) hipcc keeps the packed conversion only where nothing touches the halves afterwards; in front of relu_frags'
// 16-bit integer maximum it converted every value alone (the second operand a zero) and merged the halves with v_perm_b32: three
// instructions per pair (K2's z loop: 64 of 1,150 vector instructions)
#ifndef ENF_CVT_PK2
#define ENF_CVT_PK2 1
#endif
typedef float f32x2_cvt __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_cvt __attribute__((ext_vector_type(2)));
DEV unsigned bf16_pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cvt{a, b}, bf16x2_cvt));
}
template <bool BF16, int KB>
DEV void make_frags(Frags<BF16, KB>& F, const f32x4 (&X)[2 * KB]) {
#pragma unroll
  for (int blk = 0; blk < KB; ++blk) {
    if constexpr (BF16) {
#if ENF_CVT_PK2
      const u32x4 w = {bf16_pack2(X[2 * blk][0], X[2 * blk][1]), bf16_pack2(X[2 * blk][2], X[2 * blk][3]),
                       bf16_pack2(X[2 * blk + 1][0], X[2 * blk + 1][1]), bf16_pack2(X[2 * blk + 1][2], X[2 * blk + 1][3])};
      F.f[blk] = __builtin_bit_cast(bf16x8, w);
#else
#pragma unroll
      for (int j = 0; j < 8; ++j) F.f[blk][j] = (__bf16)X[2 * blk + (j >> 2)][j & 3];
#endif
    } else {
      F.f[2 * blk] = X[2 * blk];
      F.f[2 * blk + 1] = X[2 * blk + 1];
    }
  }
}

#ifndef ENF_ASM_LITE
#define ENF_ASM_LITE 0       // 1: asm GEMM stages keep 4 instead of 8 fragment reads in flight (register-starved kernels)
#endif
#ifndef ENF_GELU_PK
#define ENF_GELU_PK 1
#endif
#ifndef ENF_GELU_POLY
#define ENF_GELU_POLY 0
#endif
#include "enf_gemm_asm.h"
#ifndef ENF_ASM_GEMM
#define ENF_ASM_GEMM 1
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// A tile array parked in half the registers between two uses (bf16 mode: the same packing as a fragment
// set; fp32 mode: kept as is).  park() / unpark(tile): one cvt_pk per two values, one shift/and per value.
template <bool BF16, int NT> struct Parked {
  Frags<BF16, NT / 2> p;
  DEV void park(const f32x4 (&X)[NT]) { make_frags<BF16, NT / 2>(p, X); }
  DEV f32x4 get(int t) const {
    if constexpr (BF16) {
      const int blk = t >> 1, o = (t & 1) * 4;
      return f32x4{(float)p.f[blk][o], (float)p.f[blk][o + 1], (float)p.f[blk][o + 2], (float)p.f[blk][o + 3]};
    } else {
      return p.f[t];
    }
  }
};

// relu of one fp32 value as ONE integer max (negative floats, -0 included, are negative integers; positive floats pass
// through), without the canonicalising second v_max that fmaxf(x, 0) costs.
// NOT inline asm: an asm VALU instruction is invisible to hipcc's hazard recognizer, so when its input is the result of a
// compiler-scheduled MFMA no wait states are inserted and it reads a stale accumulator (gfx950 needs 7 between a
// v_mfma_f32_16x16x32_bf16 and a VALU read of its result and does not interlock: scripts/ubench/mfma_hazard.hip).  Round 1 had
// `asm("v_max_f32 %0, 0, %1")` here: harmless behind the hand-written GEMM stages (they end in s_nop 11), wrong behind the
// compiler-scheduled ones (fp32 mode; -DENF_ASM_GEMM=0) -- DESIGN.md, "K3 run-to-run deviations".
DEV float relu_f(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }

// relu on fragments.  bf16: as 16-bit integers negative floats are negative, so one v_pk_max_i16 with 0 clamps two values
// (-0 -> +0); fp32: relu_f per value.  Compiler-visible operations for the same reason as relu_f.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <bool BF16, int KB> DEV void relu_frags(Frags<BF16, KB>& F) {
  if constexpr (BF16) {
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
      const s16x8 w = __builtin_bit_cast(s16x8, F.f[blk]);
      F.f[blk] = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(w, (s16x8)0));
    }
  } else {
#pragma unroll
    for (int t = 0; t < 2 * KB; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) F.f[t][i] = relu_f(F.f[t][i]);
  }
}

// relu masks (enf_hip.h: ENF_MASK_*): one 32-bit word per lane, pair tile and relu layer; bit (4 t + i) = pre-activation
// [t][i] > 0.  relu_mask_of builds it; relu_apply_mask is the relu LINEARISED at the point the mask was taken
// (h = a where the bit is set, 0 elsewhere -- NOT max(a, 0)), which is what a finite difference of gradients needs
// to equal the almost-everywhere second derivative (relu'' = 0) instead of also counting the units that flip.
template <int NT> DEV unsigned relu_mask_of(const f32x4 (&X)[NT]) {
  static_assert(NT <= 8, "32 bits per lane");
  unsigned m = 0u;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) m |= X[t][i] > 0.f ? 1u << (4 * t + i) : 0u;
  return m;
}
template <int NT> DEV void relu_apply_mask(f32x4 (&X)[NT], unsigned m) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) X[t][i] = (m >> (4 * t + i)) & 1u ? X[t][i] : 0.f;
}
// word index of (signal b, latent z, 16-query tile, layer 0 = query RFFNet / 1 = value RFFNet, lane)
DEV size_t relu_mask_index(int b, int Z, int z, int ntiles, int tile, int layer, int lane) {
  return ((((size_t)b * Z + z) * ntiles + tile) * 2 + layer) * 64 + lane;
}

// acc[mt] += W^T[16 mt .., :] X for MTS out-tiles whose fragments start at `lds`.
// Panel order (enf_pack.hip): bf16 [mt][blk][lane] x 16 B; fp32 [mt][in-tile][lane] x 16 B.
// INIT: how the accumulators start.  INIT_ACC: the caller initialised them; INIT_ZERO: from zero (the asm stage
// feeds C = 0 to each accumulator's first MFMA: no v_mov, and the registers are not live before the stage);
// INIT_BIAS: from the per-row vector `bias` (fp32 in LDS, tile 0 of this stage), loaded inside the asm stage.
enum { INIT_ACC = 0, INIT_ZERO = 1, INIT_BIAS = 2 };
template <bool BF16, int KB, int MTS, int INIT = INIT_ACC>
DEV void gemm_stage(f32x4* acc, const Frags<BF16, KB>& F, const char* lds, int lane, const float* bias = nullptr) {
  if constexpr (BF16 && ENF_ASM_GEMM && GemmStageAsm<KB, MTS, ENF_ASM_LITE != 0>::available) {
    // hand-scheduled stage: NBUF fragment reads in flight, MFMAs round-robin over the accumulators
    using G = GemmStageAsm<KB, MTS, ENF_ASM_LITE != 0>;
    const unsigned a = (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<char*>(lds) + (lane << 4));
    if constexpr (INIT == INIT_ZERO) G::run_zero(acc, F.f, a);
    else if constexpr (INIT == INIT_BIAS) G::run_bias(acc, F.f, a, (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<float*>(bias) + 4 * (lane >> 4)));
    else G::run(acc, F.f, a);
    return;
  }
  if constexpr (INIT == INIT_ZERO) {
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else if constexpr (INIT == INIT_BIAS) {
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt) acc[mt] = *reinterpret_cast<const f32x4*>(bias + 16 * mt + 4 * (lane >> 4));
  }
#pragma unroll
  for (int mt = 0; mt < MTS; ++mt) {
    if constexpr (BF16) {
#pragma unroll
      for (int blk = 0; blk < KB; ++blk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + (((mt * KB + blk) * 64 + lane) << 4));
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, F.f[blk], acc[mt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int tin = 0; tin < 2 * KB; ++tin) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(lds + (((mt * 2 * KB + tin) * 64 + lane) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], F.f[tin][i], acc[mt], 0, 0, 0);
      }
    }
  }
}

// FLIPPED product of ONE out-tile: the same panel fragments and the same activation fragments with
// the MFMA operands swapped give (W^T X)^T, i.e. rows = this wave's columns n (register i <-> n =
// 4*quad + i) and columns = the panel's out-features (lane & 15 <-> feature 16 mt + (lane&15)).
// With the column index in registers, a sum over n (the gradient of a per-latent quantity) is a
// lane-local sum over 4 registers instead of a cross-lane reduction.
template <bool BF16, int KB>
DEV void gemm_tile_flip(f32x4& acc, const Frags<BF16, KB>& F, const char* lds, int mt, int lane) {
  if constexpr (BF16) {
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + (((mt * KB + blk) * 64 + lane) << 4));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.f[blk], a, acc, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int tin = 0; tin < 2 * KB; ++tin) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(lds + (((mt * 2 * KB + tin) * 64 + lane) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(F.f[tin][i], a[i], acc, 0, 0, 0);
    }
  }
}

// per-row constant vector (bias, u, v0 ..) -> one tile in acc layout; `vec` is fp32 in LDS or
// global; the address depends on the lane only through its quad, so the read is a broadcast.
DEV f32x4 rowvec(const float* vec, int tile, int quad) {
  return *reinterpret_cast<const f32x4*>(vec + 16 * tile + 4 * quad);
}

// sum over the four quads of a column (they hold disjoint feature rows)
#ifndef ENF_XQ_SWAP
#define ENF_XQ_SWAP 1
#endif
// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the quads of a 16-column tile), result in all four.
// ENF_XQ_SWAP: with gfx950's lane-swap instructions (pure VALU: v_permlane32_swap gives [a.lo | b.lo], [a.hi | b.hi],
// v_permlane16_swap exchanges the odd 16-lane rows of one operand with the even rows of the other) instead of two
// ds_bpermute_b32 round trips through the LDS crossbar (scripts/ubench/permlane_sum.hip: 112 vs 172 ticks per dependent sum).
// Inline asm -- the builtin loses its second result in hipcc 7.2 --, with the two wait states a VALU-written operand needs.
DEV float xquad_sum(float v) {
#if ENF_XQ_SWAP
  float a = v, b = v;
#ifndef ENF_XQ_NOP_AFTER
#define ENF_XQ_NOP_AFTER 0
#endif
#if ENF_XQ_NOP_AFTER
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float s = a + b;
  float c = s, d = s;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
#else
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  const float s = a + b;
  float c = s, d = s;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
#endif
  return c + d;
#else
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
#endif
}

// ------------------------------------------------------------------ math
// gelu, tanh approximation (jax.nn.gelu default): 0.5x(1+tanh(c(x+0.044715x^3))) = x*sigmoid(2c(..))
DEV float gelu_f(float x) {
  const float c2 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2*sqrt(2/pi)*log2(e)
  const float u = x * (c2 + (c2 * 0.044715f) * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
}
// the same on whole tiles, two values per instruction (v_pk_mul/fma/add_f32; exp and rcp stay scalar)
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEV f32x2 gelu_f2(f32x2 x) {
  const float c2 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
  const f32x2 u = x * (x * x * (c2 * 0.044715f) + c2);
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(u[0]);
  e[1] = __builtin_amdgcn_exp2f(u[1]);
  e = e + 1.0f;
  f32x2 s;
  s[0] = __builtin_amdgcn_rcpf(e[0]);
  s[1] = __builtin_amdgcn_rcpf(e[1]);
  return x * s;
}
// Transcendental-free alternative (off: ENF_GELU_POLY=0).  gelu_tanh(x) = x (0.5 + f(x)), f = 0.5 tanh(..) is odd
// and saturates: f(x) ~ xc Q(xc^2), xc = clamp(x, -4, 4), Q of degree 6 (Lawson/minimax fit, max |error| 2.8e-4).
// Measured on gfx950 (scripts/ubench/valu_rates.hip, per SIMD with two waves, v_fma_f32 = 1): v_pk_fma/mul_f32
// 1.73, v_exp/v_rcp/v_sin 2.5, v_med3 1.33 -- so 2 med3 + 9 packed ops per value pair cost what 4 transcendentals
// + 5 packed ops do: no gain, and the exact form is kept.
DEV f32x2 gelu_poly2(f32x2 x) {
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f);
  const f32x2 s = xc * xc;
  f32x2 q = s * 2.394300707e-08f + -1.664713792e-06f;
  q = q * s + 4.940474534e-05f;
  q = q * s + -8.289572461e-04f;
  q = q * s + 8.840011712e-03f;
  q = q * s + -6.464239978e-02f;
  q = q * s + 3.977454025e-01f;
  return x * (xc * q + 0.5f);
}
template <int NT, bool FAST = false> DEV void gelu_tiles(f32x4 (&X)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (FAST && ENF_GELU_POLY) {
      const f32x2 lo = gelu_poly2(f32x2{X[t][0], X[t][1]}), hi = gelu_poly2(f32x2{X[t][2], X[t][3]});
      X[t] = f32x4{lo[0], lo[1], hi[0], hi[1]};
      continue;
    }
#if ENF_GELU_PK
    const f32x2 lo = gelu_f2(f32x2{X[t][0], X[t][1]}), hi = gelu_f2(f32x2{X[t][2], X[t][3]});
    X[t] = f32x4{lo[0], lo[1], hi[0], hi[1]};
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) X[t][i] = gelu_f(X[t][i]);
#endif
  }
}
DEV f32x2 lo2(const f32x4& x) { return __builtin_shufflevector(x, x, 0, 1); }
DEV f32x2 hi2(const f32x4& x) { return __builtin_shufflevector(x, x, 2, 3); }
#ifndef ENF_GELU_DG_FMA
#define ENF_GELU_DG_FMA 1
#endif
// gelu and its derivative from ONE exp + rcp (the backward kernel needs both for the same pre-activation):
// X <- gelu(X), G <- gelu'(X), two values per instruction where the ISA has a packed form
DEV void gelu_fg2(f32x2 x, f32x2& g, f32x2& dg) {
  const float c = 0.7978845608028654f;
  const float c2 = -2.0f * c * 1.4426950408889634f;
  const f32x2 x2 = x * x;
  const f32x2 u = x * (x2 * (c2 * 0.044715f) + c2);
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(u[0]);
  e[1] = __builtin_amdgcn_exp2f(u[1]);
  e = e + 1.0f;
  f32x2 s;
  s[0] = __builtin_amdgcn_rcpf(e[0]);
  s[1] = __builtin_amdgcn_rcpf(e[1]);
  g = x * s;
#if ENF_GELU_DG_FMA
  dg = (g - g * s) * (x2 * (6.0f * c * 0.044715f) + 2.0f * c) + s;       // (g - g s: ONE packed fma; 1 - s would be two scalar subtractions)
#else
  dg = g * (1.0f - s) * (x2 * (6.0f * c * 0.044715f) + 2.0f * c) + s;
#endif
}
// the same one value at a time (no register-pair constraints for the allocator: the packed form spills in K3)
DEV void gelu_fg1(float x, float& g, float& dg) {
  const float c = 0.7978845608028654f;
  const float c2 = -2.0f * c * 1.4426950408889634f;
  const float x2 = x * x;
  const float u = x * (c2 + (c2 * 0.044715f) * x2);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
  g = x * s;
  dg = fmaf(g - g * s, fmaf(x2, 6.0f * c * 0.044715f, 2.0f * c), s);
}
// one tile: g <- gelu(x), x <- gelu'(x); the polynomial parts two values per instruction, the two transcendentals per value
DEV void gelu_fg_tile(f32x4& x, f32x4& g) {
  f32x2 g0, d0, g1, d1;
  gelu_fg2(lo2(x), g0, d0);
  gelu_fg2(hi2(x), g1, d1);
  g = f32x4{g0[0], g0[1], g1[0], g1[1]};
  x = f32x4{d0[0], d0[1], d1[0], d1[1]};
}
// in place: X <- gelu(X), returns gelu'(X) in G
template <int NT> DEV void gelu_fg_tiles(f32x4 (&X)[NT], f32x4 (&G)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x2 g0, d0, g1, d1;
    gelu_fg2(f32x2{X[t][0], X[t][1]}, g0, d0);
    gelu_fg2(f32x2{X[t][2], X[t][3]}, g1, d1);
    X[t] = f32x4{g0[0], g0[1], g1[0], g1[1]};
    G[t] = f32x4{d0[0], d0[1], d1[0], d1[1]};
  }
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
constexpr int STAGE_MAX = 32768;
constexpr unsigned NO_STAGE = 0xFFFFFFFFu;

// Stage sources are byte offsets into the packed blob.  A stage moves global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds): each wave-instruction carries 1 KB (lane-linear, which is exactly
// the fragment order the panels are packed in), no VGPR is touched, descriptor and offsets are
// scalar (cdna_hip_programming.md section 5, T8).  The DMA is issued BEFORE the MFMAs that hide its
// latency and retired by stage_wait() + the stage barrier.
DEV __amdgpu_buffer_rsrc_t make_blob_rsrc(const char* blob, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blob), 0, bytes, 0x00020000);
}

#ifndef ENF_STAGE_CONTIG
#define ENF_STAGE_CONTIG 1
#endif
// pieces I .. PPW-1 of a wave's run (the offset field wants a constant: unrolled by recursion)
template <int BYTES, bool FULL, int I, int PPW>
DEV void stage_pieces(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int voff, unsigned soff, int base) {
  if constexpr (I < PPW) {
    if (FULL || base + I * 1024 < BYTES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, I * 1024, 0);
    stage_pieces<BYTES, FULL, I + 1, PPW>(rs, dst, voff, soff, base);
  }
}
template <int BYTES, int NW = NWAVES>
DEV void stage_issue(__amdgpu_buffer_rsrc_t rs, unsigned src_off, char* dst, int wave, int lane) {
  static_assert(BYTES % 1024 == 0 && BYTES <= STAGE_MAX, "stage size");
  constexpr int PIECES = BYTES / 1024;                  // 1 KB per wave-instruction
#if ENF_STAGE_CONTIG
  // a wave copies PPW CONSECUTIVE kilobytes: the piece index then fits the instruction's 12-bit offset field (which moves the source and
  // the LDS side alike), so a stage needs ONE scalar source offset and ONE M0 value per wave.  With the pieces interleaved over the waves
  // (8 KB apart) every piece had its own pair, hipcc hoisted all of them out of the tile loop and spilled them: ~60 v_readlane_b32 per
  // K3 tile (round 3, scripts/r03_ab_vs_head.sh)
  constexpr int PPW = (PIECES + NW - 1) / NW;
  const int base = wave * (PPW * 1024);
  stage_pieces<BYTES, PIECES % NW == 0, 0, PPW>(rs, (lds_ptr_t)(dst + base), lane * 16, src_off + base, base);
#else
#pragma unroll
  for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
    const int piece = (i * NW + wave) * 1024;
    if (PIECES % NW == 0 || piece < BYTES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + piece), 16, lane * 16, src_off + piece, 0, 0);
  }
#endif
}
DEV void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Panel = (MTOUT out-tiles of 16 rows) x (KBIN in-blocks of 32) of A-operand fragments, streamed
// through a 2-slot LDS ring (slot stride STAGE_MAX) in stages of MTS out-tiles.
// `early` (wave-uniform; opt-in through first_stage<.., ANTI = true>): the upper half of the workgroup's
// waves -- the SIMD-mates of the lower half -- take each stage's barrier BEFORE its MFMAs instead of
// after them.  Every wave still meets every barrier and reads a ring slot only inside that slot's
// window, but inside a window one wave of a SIMD runs [MFMA, epilogue] while its mate runs
// [epilogue, MFMA]: the matrix pipe and the vector ALU work at the same time instead of in turns.
// A kernel that opts in must end its stage sequence with pipe_finish().
// Stage offsets with STAGE_RS2 set address the second buffer resource `rs2` (the per-latent panels of the
// z-fold forward path) instead of the weight blob `rs`; a kernel without one sets rs2 = rs.
constexpr unsigned STAGE_RS2 = 0x80000000u;
struct Pipe { __amdgpu_buffer_rsrc_t rs, rs2; int cur; int wave; bool early; };
template <int BYTES, int NW = NWAVES>
DEV void stage_issue_p(const Pipe& P, unsigned src_off, char* dst, int lane) {
#ifdef ENF_PIPE_ONE_RS       // a translation unit whose kernels stream from the weight blob only (enf_tail.hip).  With the choice below compiled
  // in, hipcc kept the whole Pipe of the tail's 2-slot kernels in scratch memory (64 bytes, no register spilled): the resource descriptor
  // came back through scratch_load + v_readfirstlane in front of every stage, behind an `s_waitcnt vmcnt(0)` that also waits for the
  // LDS-DMA stage in flight.  (The pair kernels, which do use both resources, keep theirs in scalar registers.)
  stage_issue<BYTES, NW>(P.rs, src_off, dst, P.wave, lane);
#else
  if (src_off & STAGE_RS2) stage_issue<BYTES, NW>(P.rs2, src_off & ~STAGE_RS2, dst, P.wave, lane);
  else stage_issue<BYTES, NW>(P.rs, src_off, dst, P.wave, lane);
#endif
}
DEV void stage_open(const Pipe& P) { if (P.early) { stage_wait(); __syncthreads(); } }
DEV void stage_close(Pipe& P) { if (!P.early) { stage_wait(); __syncthreads(); } P.cur ^= 1; }
DEV void pipe_finish(const Pipe& P) { if (P.early) { stage_wait(); __syncthreads(); } }
// look-ahead staging: retire this stage (its successor has landed, every wave is done with this slot), then refill this slot
template <int NEXT_BYTES, int NW>
DEV void stage_close_la(Pipe& P, char* ring, unsigned next, int lane) {
  stage_wait();
  __syncthreads();
  if (next != NO_STAGE) stage_issue_p<NEXT_BYTES, NW>(P, next, ring + P.cur * STAGE_MAX, lane);
  P.cur ^= 1;
}

template <int KBIN, int MTOUT, bool BF16> struct PanelCfg {
  static constexpr int MT_BYTES = KBIN * (BF16 ? 1024 : 2048);
  static constexpr int MTS = (STAGE_MAX / MT_BYTES) < MTOUT ? (STAGE_MAX / MT_BYTES) : MTOUT;
  static_assert(MTS >= 1 && MTOUT % MTS == 0, "panel staging");
  static constexpr int SPP = MTOUT / MTS;           // stages per panel
  static constexpr int STAGE = MTS * MT_BYTES;      // bytes per stage
  static constexpr int BYTES = MTOUT * MT_BYTES;
};

// acc[MTOUT] += panel . F.  Precondition: the panel's first stage is resident in ring[cur] and
// visible (a barrier has passed).  While stage s is multiplied, stage s+1 (or the first stage of
// `next`, NEXT_BYTES long; NO_STAGE = nothing follows) streams into the other ring slot; one
// wait + one barrier per stage publish it.  `panel` / `next` are blob byte offsets (wave-uniform).
// `active` (wave-uniform) lets a wave without work keep the staging / barrier cadence.
//
// LA ("look-ahead over the epilogue", single-stage panels only): the kernel keeps ONE stage in flight at all times instead
// of issuing it at the start of the stage before.  Precondition: this panel is resident and visible in ring[cur] AND the
// stage after it is already streaming into the other slot; `next` names the stage AFTER THAT, issued into this stage's own
// slot right behind the barrier that retires it -- so it streams under the vector-ALU epilogue that follows this call and
// under the next stage's MFMAs, not under those MFMAs alone (a D x D stage is ~1000 cycles of MFMAs for the two waves of a
// SIMD, the DMA of its successor ~3000: without this every stage that follows an epilogue waits for its panel).
// (the stages of a panel are unrolled by recursion: the stage index is a constant in every instruction's offset)
template <int SP, int KBIN, int MTOUT, bool BF16, int NEXT_BYTES, int NW, int INIT, bool LA>
DEV void panel_gemm_stages(f32x4 (&acc)[MTOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel, unsigned next,
                           bool active, int lane, const float* bias) {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  if constexpr (SP < C::SPP) {
    constexpr int sp = SP;
    stage_open(P);
    if constexpr (LA) {}
    else if constexpr (sp + 1 < C::SPP) stage_issue_p<C::STAGE, NW>(P, panel + (sp + 1) * C::STAGE, ring + (P.cur ^ 1) * STAGE_MAX, lane);
    else { if (next != NO_STAGE) stage_issue_p<NEXT_BYTES, NW>(P, next, ring + (P.cur ^ 1) * STAGE_MAX, lane); }
    if (active) gemm_stage<BF16, KBIN, C::MTS, INIT>(&acc[sp * C::MTS], F, ring + P.cur * STAGE_MAX, lane, bias + 16 * sp * C::MTS);
    else if constexpr (INIT != INIT_ACC) {        // a wave that only keeps the barrier cadence still gets defined values
#pragma unroll
      for (int mt = 0; mt < C::MTS; ++mt)
        acc[sp * C::MTS + mt] = INIT == INIT_BIAS ? *reinterpret_cast<const f32x4*>(bias + 16 * (sp * C::MTS + mt) + 4 * (lane >> 4))
                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (LA) stage_close_la<NEXT_BYTES, NW>(P, ring, next, lane);
    else stage_close(P);
    panel_gemm_stages<SP + 1, KBIN, MTOUT, BF16, NEXT_BYTES, NW, INIT, LA>(acc, F, P, ring, panel, next, active, lane, bias);
  }
}
template <int KBIN, int MTOUT, bool BF16, int NEXT_BYTES, int NW = NWAVES, int INIT = INIT_ACC, bool LA = false>
DEV void panel_gemm(f32x4 (&acc)[MTOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel, unsigned next,
                    bool active, int lane, const float* bias = nullptr) {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  static_assert(!LA || C::SPP == 1, "look-ahead staging: single-stage panels");
  panel_gemm_stages<0, KBIN, MTOUT, BF16, NEXT_BYTES, NW, INIT, LA>(acc, F, P, ring, panel, next, active, lane, bias);
}

// panel_gemm that additionally hands every out-tile's FLIPPED product to `flip(tile, acc)`
// (acc starts at flip_init(tile)); TRANS = false skips the transposed product.
template <int KBIN, int MTOUT, bool BF16, int NEXT_BYTES, int NW, bool TRANS, int INIT = INIT_ACC, bool LA = false, typename InitFn, typename FlipFn>
DEV void panel_gemm_flip(f32x4 (&acc)[TRANS ? MTOUT : 1], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned panel,
                         unsigned next, int lane, InitFn&& flip_init, FlipFn&& flip, const float* bias = nullptr) {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  static_assert(!LA || C::SPP == 1, "look-ahead staging: single-stage panels");
#pragma unroll
  for (int sp = 0; sp < C::SPP; ++sp) {
    stage_open(P);
    if constexpr (LA) {}
    else if (sp + 1 < C::SPP) stage_issue_p<C::STAGE, NW>(P, panel + (sp + 1) * C::STAGE, ring + (P.cur ^ 1) * STAGE_MAX, lane);
    else if (next != NO_STAGE) stage_issue_p<NEXT_BYTES, NW>(P, next, ring + (P.cur ^ 1) * STAGE_MAX, lane);
    const char* slot = ring + P.cur * STAGE_MAX;
    if constexpr (TRANS && BF16 && ENF_ASM_GEMM && GemmStageAsm<KBIN, C::MTS, ENF_ASM_LITE != 0>::available) {
      // one fragment read feeds both products (hand-scheduled stage, enf_gemm_asm.h)
      f32x4 af[C::MTS];
#pragma unroll
      for (int mt = 0; mt < C::MTS; ++mt) af[mt] = flip_init(sp * C::MTS + mt);
      using G = GemmStageAsm<KBIN, C::MTS, ENF_ASM_LITE != 0>;
      const unsigned a = (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<char*>(slot) + (lane << 4));
      if constexpr (INIT == INIT_ZERO) G::run_both_zero(&acc[sp * C::MTS], af, F.f, a);
      else if constexpr (INIT == INIT_BIAS)
        G::run_both_bias(&acc[sp * C::MTS], af, F.f, a, (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<float*>(bias) + 16 * sp * C::MTS + 4 * (lane >> 4)));
      else G::run_both(&acc[sp * C::MTS], af, F.f, a);
#pragma unroll
      for (int mt = 0; mt < C::MTS; ++mt) flip(sp * C::MTS + mt, af[mt]);
    } else if constexpr (!TRANS && BF16 && ENF_ASM_GEMM && GemmStageAsm<KBIN, C::MTS, ENF_ASM_LITE != 0>::available) {
      // flipped product only
      using G = GemmStageAsm<KBIN, C::MTS, ENF_ASM_LITE != 0>;
      const unsigned a = (unsigned)(uintptr_t)(lds_ptr_t)(const_cast<char*>(slot) + (lane << 4));
      f32x4 af[C::MTS];
      if constexpr (INIT == INIT_ZERO) {
        G::run_flip_zero(af, F.f, a);
      } else {
#pragma unroll
        for (int mt = 0; mt < C::MTS; ++mt) af[mt] = flip_init(sp * C::MTS + mt);
        G::run_flip(af, F.f, a);
      }
#pragma unroll
      for (int mt = 0; mt < C::MTS; ++mt) flip(sp * C::MTS + mt, af[mt]);
    } else {
      if constexpr (TRANS) gemm_stage<BF16, KBIN, C::MTS, INIT>(&acc[sp * C::MTS], F, slot, lane, bias + 16 * sp * C::MTS);
#pragma unroll
      for (int mt = 0; mt < C::MTS; ++mt) {
        f32x4 af = flip_init(sp * C::MTS + mt);
        gemm_tile_flip<BF16, KBIN>(af, F, slot, mt, lane);
        flip(sp * C::MTS + mt, af);
      }
    }
    if constexpr (LA) stage_close_la<NEXT_BYTES, NW>(P, ring, next, lane);
    else stage_close(P);
  }
}


// ---- "A3": antiphase over a THREE-slot ring (single-stage panels).  The upper half of the workgroup's waves -- the SIMD-mates of
// the lower half -- take each stage's barrier BEFORE its MFMAs, the lower half after them, so that on every SIMD one wave multiplies
// while the other runs its vector epilogue.  Barrier b (the same instance for all waves) therefore completes when the late waves
// have finished G_b and the early ones V_{b-1}: slot b % 3 is still being read by the early waves, panel b + 1 must be (and is:
// issued behind barrier b - 1, waited for by every wave before it arrives) resident for the late waves' next G, and slot
// (b + 2) % 3 -- last read for G_{b-1}, which every wave has behind it -- is free: panel b + 2 is issued into it behind the barrier.
// Two slots cannot do this (the refill of a slot would race the early waves' reads of it); both halves meet every barrier once.
template <int BYTES0, int BYTES1, int NW = NWAVES>
DEV void first_stage_a3(Pipe& P, char* ring, unsigned panel0, unsigned panel1, int wave, int lane) {
  P.cur = 0;
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  P.early = P.wave >= NW / 2;
  stage_issue_p<BYTES0, NW>(P, panel0, ring, lane);
  stage_issue_p<BYTES1, NW>(P, panel1, ring + STAGE_MAX, lane);
  stage_wait();
  __syncthreads();
}
template <int KBIN, int MTOUT, bool BF16, int NEXT_BYTES, int NW = NWAVES, int INIT = INIT_ACC>
DEV void panel_gemm_a3(f32x4 (&acc)[MTOUT], const Frags<BF16, KBIN>& F, Pipe& P, char* ring, unsigned next2, bool active, int lane,
                       const float* bias = nullptr) {
  using C = PanelCfg<KBIN, MTOUT, BF16>;
  static_assert(C::SPP == 1, "A3 staging: single-stage panels");
  char* refill = ring + ((P.cur + 2) % 3) * STAGE_MAX;
  if (P.early) {
    stage_wait();
    __syncthreads();
    if (next2 != NO_STAGE) stage_issue_p<NEXT_BYTES, NW>(P, next2, refill, lane);
  }
  if (active) gemm_stage<BF16, KBIN, C::MTS, INIT>(acc, F, ring + P.cur * STAGE_MAX, lane, bias);
  else if constexpr (INIT != INIT_ACC) {
#pragma unroll
    for (int mt = 0; mt < C::MTS; ++mt)
      acc[mt] = INIT == INIT_BIAS ? *reinterpret_cast<const f32x4*>(bias + 16 * mt + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (!P.early) {
    stage_wait();
    __syncthreads();
    if (next2 != NO_STAGE) stage_issue_p<NEXT_BYTES, NW>(P, next2, refill, lane);
  }
  P.cur = (P.cur + 1) % 3;
}

template <int BYTES, int NW = NWAVES, bool ANTI = false>
DEV void first_stage(Pipe& P, char* ring, unsigned panel, int wave, int lane) {
  P.cur = 0;
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  P.early = ANTI && P.wave >= NW / 2;
  stage_issue_p<BYTES, NW>(P, panel, ring, lane);
  if constexpr (ANTI) __syncthreads();      // publishes the kernel's LDS constants to the early waves
  if (!P.early) { stage_wait(); __syncthreads(); }
}

// Per-lane sums over a lane's NT x 4 values, two at a time: even / odd running sums in one register pair each, so a sum costs one
// v_pk_add_f32 / v_pk_fma_f32 per TWO elements (plain operands: no op_sel, no neg).  Written with a single accumulator the additions are a
// serial chain the compiler may not reassociate: 64 vector instructions per LayerNorm instead of 34 -- 6 % of K3's, 8 % of K2's.
#ifndef ENF_PK_SUMS
#define ENF_PK_SUMS 1
#endif
// s = sum x, q = sum x^2
template <int NT> DEV void tiles_sum_sq(const f32x4 (&X)[NT], float& s, float& q) {
#if ENF_PK_SUMS
  f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x2 a = lo2(X[t]), b = hi2(X[t]);
    s2 += a; q2 = __builtin_elementwise_fma(a, a, q2);
    s2 += b; q2 = __builtin_elementwise_fma(b, b, q2);
  }
  s = s2[0] + s2[1]; q = q2[0] + q2[1];
#else
  s = 0.f; q = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += X[t][i]; q = fmaf(X[t][i], X[t][i], q); }
#endif
}
// s = sum a, d = sum a b      (`get(t)` yields tile t of b: a register array or parked fragments)
template <int NT, typename GetB> DEV void tiles_sum_dot(const f32x4 (&A)[NT], GetB get, float& s, float& d) {
#if ENF_PK_SUMS
  f32x2 s2 = {0.f, 0.f}, d2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 bt = get(t);
    s2 += lo2(A[t]); d2 = __builtin_elementwise_fma(lo2(A[t]), lo2(bt), d2);
    s2 += hi2(A[t]); d2 = __builtin_elementwise_fma(hi2(A[t]), hi2(bt), d2);
  }
  s = s2[0] + s2[1]; d = d2[0] + d2[1];
#else
  s = 0.f; d = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 bt = get(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += A[t][i]; d = fmaf(A[t][i], bt[i], d); }
  }
#endif
}

// d = sum a b
template <int NT, typename GetA, typename GetB> DEV float tiles_dot(GetA geta, GetB getb) {
#if ENF_PK_SUMS
  f32x2 d2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 at = geta(t), bt = getb(t);
    d2 = __builtin_elementwise_fma(lo2(at), lo2(bt), d2);
    d2 = __builtin_elementwise_fma(hi2(at), hi2(bt), d2);
  }
  return d2[0] + d2[1];
#else
  float d = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 at = geta(t), bt = getb(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) d = fmaf(at[i], bt[i], d);
  }
  return d;
#endif
}

// LayerNorm statistics over the NT*16 features of this lane's column: biased variance, eps 1e-6,
// one pass (E[x^2] - E[x]^2, flax.linen.LayerNorm's default use_fast_variance=True)
template <int NT> DEV void ln_stats(const f32x4 (&X)[NT], float& mu, float& rstd, float inv_n = 1.0f / (16 * NT)) {
  float s, q;
  tiles_sum_sq<NT>(X, s, q);
  mu = xquad_sum(s) * inv_n;          // inv_n = 1 / (number of real features): zero padding adds nothing to the sums
  const float ex2 = xquad_sum(q) * inv_n;
  rstd = rsqrtf(fmaxf(ex2 - mu * mu, 0.f) + 1e-6f);
}

// x <- (x - mu) * rstd, as ONE v_fma_f32 per element written in asm.  Left as a plain loop, hipcc's SLP vectoriser turns it
// into v_pk_add_f32 (op_sel broadcast of mu, negated) + v_pk_mul_f32, and on gfx950 the LOW half of such a v_pk_add_f32 was
// caught losing its broadcast operand in lanes 48-63 -- the mean is not subtracted from one feature of the 16 queries of a
// tile -- in the younger wave of a SIMD, about once per 10^5 executions and only after another kernel had left the CU in a
// particular state: the long-standing "K3 run-to-run deviations" (scripts/k3_race/store_probe.py finds it in
// seconds, 47 of 47 events with this signature, 0 in 4000 runs of a build without SLP packing).  The asm is opaque to the
// vectoriser; the leading s_nop covers a transcendental producer (v_rsq_f32) the compiler cannot see being consumed here.
#ifndef ENF_LN_APPLY_ASM
#define ENF_LN_APPLY_ASM 2
#endif
#ifdef ENF_LN_FORM           // investigation builds: one explicit packed-operand form (scripts/k3_race/ln_forms.h)
#include "../../scripts/k3_race/ln_forms.h"
#endif
template <int NT> DEV void ln_apply(f32x4 (&X)[NT], float mu, float rstd) {
#ifdef ENF_LN_FORM
  ln_apply_form<NT>(X, mu, rstd);
#elif ENF_LN_APPLY_ASM == 2
  // as below, with the wait states where the hazards are instead of one s_nop per element: ONE in front (rstd comes out of
  // v_rsq_f32, and a transcendental's result needs a wait state before a VALU read the compiler cannot see), and the two of
  // VALU -> MFMA operand behind the LAST fma, tied to every tile by data dependence (an asm without operands orders only
  // against other volatile asm: a compiler-scheduled MFMA reading X -- fp32 mode, where the fragments ARE these registers --
  // could otherwise be placed above it)
  float nmr = -mu * rstd;
  asm volatile("s_nop 0" : "+v"(rstd), "+v"(nmr));
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float y;
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(X[t][i]), "v"(rstd), "v"(nmr));
      X[t][i] = y;
    }
  if constexpr (NT == 8)
    asm volatile("s_nop 1" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]));
  else if constexpr (NT == 4)
    asm volatile("s_nop 1" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]));
  else {
    static_assert(NT == 16, "ln_apply: tile counts 4, 8, 16");
    asm volatile("s_nop 1" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]),
                 "+v"(X[8]), "+v"(X[9]), "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]));
  }
#elif ENF_LN_APPLY_ASM
  const float nmr = -mu * rstd;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float y;
      asm volatile("s_nop 0\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(X[t][i]), "v"(rstd), "v"(nmr));
      X[t][i] = y;
    }
  // the results may feed a compiler-scheduled MFMA directly (fp32 mode: the fragments ARE these registers), and the compiler
  // does not know a vector instruction wrote them: the two wait states of VALU -> MFMA operand (scripts/ubench), behind the
  // last fma (volatile statements keep their order)
  asm volatile("s_nop 1");
#else
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) X[t][i] = (X[t][i] - mu) * rstd;
#endif
}

// ------------------------------------------------------------------ invariants + window
// xq: query coordinate (dx <= 3) of this lane's column; pz: latent pose row from the latent
// table: periodic/rel/abs/norm -> (p0,p1,p2,-); ponita -> (px,py,cos t,sin t);
// sphere -> (phi, theta, sin theta, cos theta).  sx/cx = sin/cos(theta_x) (sphere only).
struct QueryPt { float x0, x1, x2, sx, cx; };

DEV QueryPt make_query(float x0, float x1, float x2, int inv_id) {
  QueryPt q;
  q.x0 = x0; q.x1 = x1; q.x2 = x2;
  q.sx = 0.f; q.cx = 0.f;
  if (inv_id == ENF_INV_LATITUDE_PERIODIC || inv_id == ENF_INV_POLAR_PERIODIC || enf_inv_has_phase(inv_id)) { q.sx = sinf(q.x1); q.cx = cosf(q.x1); }
  else if (inv_id == ENF_INV_PONITA_FULL) { q.sx = sinf(q.x2); q.cx = cosf(q.x2); }      // the query's own orientation
  return q;
}
DEV QueryPt load_query(const float* xp, int dx, int inv_id) {
  return make_query(xp[0], dx > 1 ? xp[1] : 0.f, dx > 2 ? xp[2] : 0.f, inv_id);
}

// `ext`: the latent's 16-float extension (LDS, wave-uniform): ball -> the rotation matrix R (row-major)
template <bool FAST>
DEV void pair_invariant(int inv_id, int dx, const QueryPt& q, const f32x4& pz, float wcoef, int use_window,
                        float (&inv)[4], float& win, const float* ext = nullptr) {
  inv[0] = inv[1] = inv[2] = inv[3] = 0.f;
  win = 0.f;
  switch (inv_id) {
    case ENF_INV_BALL:                         // ball.py:54-96: [R x^, r_x | r_p -> phase]; window ball.py:36-52
    case ENF_INV_BALL_LAT: {                   // ball_lat.py:66-88: [th_x, cos dphi, sin dphi, r_x | th_p, r_p -> phase]
      const float dphi = (q.x0 - pz[0]) * 0.15915494309189535f;
      const float cd = cos_rev<FAST>(dphi), sd = sin_rev<FAST>(dphi);
      if (inv_id == ENF_INV_BALL) {
        const float xr = q.x0 * 0.15915494309189535f;
        const float xh0 = q.sx * cos_rev<FAST>(xr), xh1 = q.sx * sin_rev<FAST>(xr), xh2 = q.cx;     // unit vector of the query
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(ext), r1 = *reinterpret_cast<const f32x4*>(ext + 4);
        const float r8 = ext[8];
        inv[0] = r0[0] * xh0 + r0[1] * xh1 + r0[2] * xh2;
        inv[1] = r0[3] * xh0 + r1[0] * xh1 + r1[1] * xh2;
        inv[2] = r1[2] * xh0 + r1[3] * xh1 + r8 * xh2;
        inv[3] = q.x2;
      } else { inv[0] = q.x1; inv[1] = cd; inv[2] = sd; inv[3] = q.x2; }
      if (use_window) {
        const float dot = q.sx * pz[2] * cd + q.cx * pz[3];
        const float dc = fminf(fmaxf(dot, -1.f + 1e-6f), 1.f - 1e-6f);
        const float ang = acosf(dc);
        win = __expf(-ang * ang * wcoef);
      }
    } break;
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
    case ENF_INV_PONITA_FULL:                  // ponita.py:80-92: the two below and the orientations' inner product
    case ENF_INV_PONITA: {                     // ponita.py:30-44; window _base_invariant.py:25-33
      const float r0 = q.x0 - pz[0], r1 = q.x1 - pz[1];
      inv[0] = r0 * pz[2] + r1 * pz[3];
      inv[1] = -r0 * pz[3] + r1 * pz[2];
      if (inv_id == ENF_INV_PONITA_FULL) inv[2] = q.cx * pz[2] + q.sx * pz[3];
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
