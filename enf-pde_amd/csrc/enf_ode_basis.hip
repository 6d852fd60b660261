// enf_ode_basis.hip -- the kernel basis of the latent ODE's message passing, fused.
//
// PonitaGen (experiments/fitting/ode_models/ponita_ode_g.py:128-131, 158-160) maps the invariants of every latent pair to
//     kb = gelu(gelu(poly(inv) W1 + b1) W3 + b3),      poly = [x, x(x)x, x(x)x(x)x, x(x)x(x)x(x)x]   (:15-26, degree 3)
// The reference (and the unfused path of fitting/ode_models) materialises poly(inv): (B Z^2, F) with F = I + I^2 + I^3 + I^4
// (340 for I = 4: 89 MB per evaluation at B = 16, Z = 64), then two GEMMs and, backwards, five more plus the product rule.
// Here one kernel each way keeps everything between `inv` and `kb` in registers (fp32 16x16x4 MFMA):
//   forward : features are GENERATED as the MFMA's B operand.  The sum over the MFMA's K index is order-free, so the F
//             features are re-ordered to  16 S + 4 q + j  <->  x_q * mid_S * x_j  (q: leading index = the lane's quad,
//             mid_S in {1, x_a, x_a x_b}, j: trailing index = the K-step): a lane forms its four B values of a super-step
//             with five multiplies, and its four A values are ONE 16-byte load from the packed, transposed W1.
//   backward: recomputes the forward per 16-pair tile, then d pre2, d h1, d pre1, d features -> d inv by the product rule
//             in the same (q, mid, j) order; the weight gradients contract over the PAIR axis, so d pre1 / h1 / d pre2 go
//             through LDS once (transposed: pairs along K) and every wave of the 4-wave workgroup accumulates the output
//             tiles IT owns over all 64 pairs of the workgroup's tile -- in registers over the whole launch (one wave per
//             SIMD: the 176 + 32 accumulator registers of the 128 / 64 / I = 4 shape live in the AGPR half of a 512-register
//             budget; weight operands are fetched one super-step ahead); one partial per workgroup, summed in a fixed
//             order by the reduce kernel (bitwise reproducible, no atomics).
// I <= 4 (smaller I: zero-padded components meet zero weight rows), degree 3, hidden / basis widths multiples of 16.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_launch.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define OB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int I> struct ObFeat {
  static constexpr int NS = 1 + I + I * I;       // super-steps of 16 packed features (mid = 1 | x_a | x_a x_b)
  static constexpr int FP = 16 * (NS + 1);       // + one tile: the degree-1 features x_c at 16 NS + c, a constant 1 at 16 NS + 4
  static constexpr int F = I + I * I + I * I * I + I * I * I * I;
};

// packed feature index -> row of the reference's W1 (-1: zero padding, -2: the constant-1 feature = the bias b1)
__host__ __device__ inline int ob_orig_row(int I, int fp) {
  const int NS = 1 + I + I * I, S = fp >> 4, rem = fp & 15;
  if (S == NS) return rem < I ? rem : (rem == 4 ? -2 : -1);
  const int q = rem >> 2, j = rem & 3;
  if (q >= I || j >= I) return -1;
  if (S == 0) return I + q * I + j;
  if (S <= I) return I + I * I + (q * I + (S - 1)) * I + j;
  const int ab = S - 1 - I;                       // a = ab / I, b = ab % I
  return I + I * I + I * I * I + ((q * I + ab / I) * I + ab % I) * I + j;
}

__device__ __forceinline__ float ob_pick(const float (&x)[4], int i) {
  float v = x[0];
  v = i == 1 ? x[1] : v;
  v = i == 2 ? x[2] : v;
  v = i == 3 ? x[3] : v;
  return v;
}
__device__ __forceinline__ float ob_pick4(const f32x4& x, int i) {
  float v = x[0];
  v = i == 1 ? x[1] : v;
  v = i == 2 ? x[2] : v;
  v = i == 3 ? x[3] : v;
  return v;
}
// mid_S for a (wave-uniform, run-time) super-step, BRANCH-FREE: as indices (ia, ib) into {1, x_0, .., x_3} (scalar selects).
// With `if (S == 0) .. else if (S <= I) ..` the loop body becomes several basic blocks and every s_waitcnt at their joins is
// vmcnt(0): the operands fetched ahead for the NEXT super-steps were waited for as well.
template <int I> __device__ __forceinline__ void ob_mid_idx(int S, int& ia, int& ib) {
  const int ab = S - 1 - I;
  ia = S == 0 ? 0 : (S <= I ? S : ab / I + 1);
  ib = S <= I ? 0 : ab % I + 1;
}
__device__ __forceinline__ float ob_pick1(const float (&x)[4], int i) {   // {1, x_0, .., x_3}[i]
  float v = 1.f;
  v = i == 1 ? x[0] : v;
  v = i == 2 ? x[1] : v;
  v = i == 3 ? x[2] : v;
  v = i == 4 ? x[3] : v;
  return v;
}
template <int I> __device__ __forceinline__ float ob_mid(const float (&x)[4], int S) {
  int ia, ib;
  ob_mid_idx<I>(S, ia, ib);
  return ob_pick1(x, ia) * ob_pick1(x, ib);
}
template <int I> __device__ __forceinline__ float ob_mid4(const f32x4& x, int S) {
  if (S == 0) return 1.f;
  if (S <= I) return ob_pick4(x, S - 1);
  const int ab = S - 1 - I;
  return ob_pick4(x, ab / I) * ob_pick4(x, ab % I);
}

// gelu, tanh form (flax nn.gelu default), and its derivative
__device__ __forceinline__ float ob_tanh(float u) {
  const float e = __expf(2.f * u);                 // e = inf -> 1, e = 0 -> -1
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float ob_gelu(float x) {
  return 0.5f * x * (1.f + ob_tanh(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ void ob_gelu_g(float x, float& y, float& dy) {
  const float x2 = x * x, t = ob_tanh(0.7978845608028654f * (x + 0.044715f * x * x2));
  y = 0.5f * x * (1.f + t);
  dy = 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 0.134145f * x2);
}

struct ObArgs {
  const float* inv; long P;                        // (P, I)
  const float* W1T;                                // (H1, FP): packed features along the row
  const float* W1P;                                // (FP, H1)
  const float* b1;
  const float* W3T;                                // (J, H1)
  const float* W3;                                 // (H1, J): the reference's tensor
  const float* b3;
  float* kb;                                       // forward: (P, J)
  const float* dkb;                                // backward: (P, J)
  float* dinv;                                     // (P, I)
  float* part;                                     // per workgroup: d W1T (H1, FP) | d W3T (J, H1) | d b3 (4 waves, J)
  int ntiles;
};

// --------------------------------------------------------------------------------------------------------------- pack
struct ObPackArgs { const float* W1; const float* W3; float* W1T; float* W1P; float* W3T; int I, H1, J, FP; };
__global__ __launch_bounds__(256) void enf_ode_basis_pack_kernel(ObPackArgs A) {
  const int id = blockIdx.x * 256 + threadIdx.x, n1 = A.FP * A.H1;
  if (id < n1) {
    const int fp = id / A.H1, h = id % A.H1, row = ob_orig_row(A.I, fp);
    const float v = row >= 0 ? A.W1[(size_t)row * A.H1 + h] : 0.f;
    A.W1P[id] = v;
    A.W1T[(size_t)h * A.FP + fp] = v;
  } else if (id < n1 + A.J * A.H1) {
    const int k = id - n1, j = k / A.H1, h = k % A.H1;
    A.W3T[k] = A.W3[(size_t)h * A.J + j];
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
// 4 waves; a wave owns PT tiles of 16 pairs (the pairs are the MFMA's columns) and sweeps all H1 hidden rows, so a weight
// operand loaded once meets PT x H1T independent accumulators.
template <int I, int H1T, int JT, int PT>
__global__ __launch_bounds__(256) void enf_ode_basis_fwd_kernel(ObArgs A) {
  constexpr int NS = ObFeat<I>::NS, FP = ObFeat<I>::FP, H1 = 16 * H1T, J = 16 * JT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
  const long p0 = ((long)blockIdx.x * 4 + wave) * (16 * PT);
  if (p0 >= A.P) return;                                            // wave-uniform, no barrier in this kernel
  float x[PT][4], xl[PT];
  bool ok[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const long p = p0 + 16 * pt + col;
    ok[pt] = p < A.P;
    const float* xp = A.inv + (ok[pt] ? p : A.P - 1) * I;
#pragma unroll
    for (int c = 0; c < 4; ++c) x[pt][c] = c < I ? xp[c] : 0.f;
    xl[pt] = ob_pick(x[pt], quad);
  }
  f32x4 acc[PT][H1T];
#pragma unroll
  for (int t = 0; t < H1T; ++t) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(A.b1 + 16 * t + 4 * quad);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[pt][t] = b;
  }
  const float* w1 = A.W1T + (size_t)col * FP + 4 * quad;
  for (int S = 0; S < NS; ++S) {
    f32x4 wa[H1T], bv[PT];
#pragma unroll
    for (int t = 0; t < H1T; ++t) wa[t] = *reinterpret_cast<const f32x4*>(w1 + (size_t)16 * t * FP + 16 * S);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const float m = ob_mid<I>(x[pt], S) * xl[pt];
      bv[pt] = f32x4{m * x[pt][0], m * x[pt][1], m * x[pt][2], m * x[pt][3]};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < H1T; ++t)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[pt][t] = OB_MFMA(wa[t][r], bv[pt][r], acc[pt][t]);
  }
  {                                                                  // degree 1: K = the component index = the quad
    const float* w1d = A.W1T + (size_t)col * FP + 16 * NS + quad;
#pragma unroll
    for (int t = 0; t < H1T; ++t) {
      const float wa = w1d[(size_t)16 * t * FP];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[pt][t] = OB_MFMA(wa, xl[pt], acc[pt][t]);
    }
  }
  f32x4 acc2[PT][JT];
#pragma unroll
  for (int t2 = 0; t2 < JT; ++t2) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(A.b3 + 16 * t2 + 4 * quad);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc2[pt][t2] = b;
  }
  const float* w3 = A.W3T + (size_t)col * H1 + 4 * quad;
#pragma unroll
  for (int t1 = 0; t1 < H1T; ++t1) {
    f32x4 wa[JT], h[PT];
#pragma unroll
    for (int t2 = 0; t2 < JT; ++t2) wa[t2] = *reinterpret_cast<const f32x4*>(w3 + (size_t)16 * t2 * H1 + 16 * t1);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[pt][r] = ob_gelu(acc[pt][t1][r]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t2 = 0; t2 < JT; ++t2)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc2[pt][t2] = OB_MFMA(wa[t2][r], h[pt][r], acc2[pt][t2]);
  }
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    if (!ok[pt]) continue;
    float* o = A.kb + (size_t)(p0 + 16 * pt + col) * J + 4 * quad;
#pragma unroll
    for (int t2 = 0; t2 < JT; ++t2) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = ob_gelu(acc2[pt][t2][r]);
      *reinterpret_cast<f32x4*>(o + 16 * t2) = v;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
// LDS operand buffers, "pairs along K": feature f of pair (tile pw, 4 kq + e) at ((pw 4 + kq) GS + 4 f + e), GS = 4 NF + 4
// (the + 4 spreads a writing wave's 64 lanes over the 64 banks; rows stay 16-byte aligned for the float4 operand reads).
constexpr int OB_NW = 4;                           // waves per backward workgroup; its pair tile is 16 OB_NW pairs
template <int NF> struct ObLds { static constexpr int GS = 4 * NF + 4, WORDS = 4 * OB_NW * GS; };

template <int I, int H1T, int JT> struct ObBwd {
  static constexpr int NS = ObFeat<I>::NS, FP = ObFeat<I>::FP, H1 = 16 * H1T, J = 16 * JT;
  static constexpr int NT1 = H1T * (NS + 1), NW1 = (NT1 + OB_NW - 1) / OB_NW;   // d W1T tiles (16 h x 16 packed features), per wave
  static constexpr int NT3 = JT * H1T, NW3 = (NT3 + OB_NW - 1) / OB_NW;         // d W3T tiles (16 j x 16 h)
  static constexpr int LDS_BASE = 4 * (ObLds<H1>::WORDS + ObLds<J>::WORDS + 16 * OB_NW * 4);
  // gelu'(pre1) waits for d h1 across three phases: parked in LDS (each lane its own slots) when that still fits 160 KB,
  // which leaves the registers to the two-step weight prefetch
  static constexpr bool STASH = LDS_BASE + 4 * H1 * 16 * OB_NW <= 160 * 1024;
  static constexpr int LDS_BYTES = LDS_BASE + (STASH ? 4 * H1 * 16 * OB_NW : 0);
  static constexpr int PART = H1 * FP + J * H1 + OB_NW * J;            // floats per workgroup partial
  static constexpr int HPW = H1T % OB_NW == 0 ? H1T / OB_NW : 0;       // h tiles of d W1T a wave owns (0: generic tile split)
};

template <int I, int H1T, int JT>
__global__ __launch_bounds__(64 * OB_NW) void enf_ode_basis_bwd_kernel(ObArgs A) {
  using C = ObBwd<I, H1T, JT>;
  constexpr int NS = C::NS, FP = C::FP, H1 = C::H1, J = C::J, NW1 = C::NW1, NW3 = C::NW3;
  constexpr int GSH = ObLds<H1>::GS, GSJ = ObLds<J>::GS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sH = reinterpret_cast<float*>(smem);                  // h1, later d pre1
  float* sJ = sH + ObLds<H1>::WORDS;                           // d pre2
  float* sX = sJ + ObLds<J>::WORDS;                            // inv of the tile's pairs, 4 floats each
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, quad = lane >> 4;
  float* sG = sX + 16 * OB_NW * 4 + tid;                        // + 64 OB_NW (4 t + r): this lane's gelu'(pre1)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: MFMAs sit behind wave-dependent branches

  f32x4 accW1[NW1], accW3[NW3], db3[JT];
#pragma unroll
  for (int i = 0; i < NW1; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NW3; ++i) accW3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < JT; ++t) db3[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's slots in the LDS operand buffers: writer (pair = col, features 4 quad + r of tile t), reader (row col, group quad)
  float* wH = sH + (wave * 4 + (col >> 2)) * GSH + 16 * quad + (col & 3);      // + 64 t + 4 r
  float* wJ = sJ + (wave * 4 + (col >> 2)) * GSJ + 16 * quad + (col & 3);
  const float* rH = sH + quad * GSH + 4 * col;                                 // + pw 4 GSH + 64 tile
  const float* rJ = sJ + quad * GSJ + 4 * col;

  for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    const long p = (long)tile * (16 * OB_NW) + wave * 16 + col;
    const bool ok = p < A.P;
    const long pc = ok ? p : A.P - 1;
    float x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = c < I ? A.inv[pc * I + c] : 0.f;
    const float xl = ob_pick(x, quad);
    if (quad == 0) *reinterpret_cast<f32x4*>(sX + (wave * 16 + col) * 4) = f32x4{x[0], x[1], x[2], x[3]};

    // ---- forward again: pre1 -> h1, gelu'(pre1)
    f32x4 acc[H1T], g1[C::STASH ? 1 : H1T];
#pragma unroll
    for (int t = 0; t < H1T; ++t) acc[t] = *reinterpret_cast<const f32x4*>(A.b1 + 16 * t + 4 * quad);
    {
      const float* w1 = A.W1T + (size_t)col * FP + 4 * quad;
      // weight operands come from L2 (the packed W1 is 180 KB): fetched TWO super-steps ahead through a ring of three
      // register buffers (one super-step = 32 MFMAs = ~1000 cycles, about one L2 round trip for the only wave of the SIMD)
      f32x4 wr0[H1T], wr1[H1T], wr2[H1T];
      auto ld = [&](f32x4 (&w)[H1T], int S) {
        const int Sc = S < NS ? S : NS - 1;
#pragma unroll
        for (int t = 0; t < H1T; ++t) w[t] = *reinterpret_cast<const f32x4*>(w1 + (size_t)16 * t * FP + 16 * Sc);
      };
      auto mm = [&](const f32x4 (&w)[H1T], int S) {
        const float m = ob_mid<I>(x, S) * xl;
        const f32x4 bv = f32x4{m * x[0], m * x[1], m * x[2], m * x[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < H1T; ++t) acc[t] = OB_MFMA(w[t][r], bv[r], acc[t]);
      };
#ifndef OB_SKIP_GEMM1   // (timing probes only: scripts/probe_ode_basis.py)
      ld(wr0, 0);
      ld(wr1, 1);
      int S = 0;
#pragma nounroll
      for (; S + 3 <= NS; S += 3) {
        ld(wr2, S + 2); mm(wr0, S);
        ld(wr0, S + 3); mm(wr1, S + 1);
        ld(wr1, S + 4); mm(wr2, S + 2);
      }
      if (S < NS) mm(wr0, S);
      if (S + 1 < NS) mm(wr1, S + 1);
#endif
      const float* w1d = A.W1T + (size_t)col * FP + 16 * NS + quad;
#pragma unroll
      for (int t = 0; t < H1T; ++t) acc[t] = OB_MFMA(w1d[(size_t)16 * t * FP], xl, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < H1T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y, dy;
        ob_gelu_g(acc[t][r], y, dy);
        acc[t][r] = y;
        if constexpr (C::STASH) sG[64 * OB_NW * (4 * t + r)] = dy;
        else g1[t][r] = dy;
        wH[64 * t + 4 * r] = y;
      }
    // ---- pre2, d pre2 = d kb * gelu'(pre2).  (Rolled loops with the B operand read back from this lane's own LDS
    //      slots: unrolled, the scheduler hoists every weight load of the phase and spills the persistent accumulators.)
    f32x4 dp2[JT];
#pragma unroll
    for (int t2 = 0; t2 < JT; ++t2) dp2[t2] = *reinterpret_cast<const f32x4*>(A.b3 + 16 * t2 + 4 * quad);
    {
      const float* w3 = A.W3T + (size_t)col * H1 + 4 * quad;
      f32x4 wa[JT], wn[JT];
#pragma unroll
      for (int t2 = 0; t2 < JT; ++t2) wa[t2] = *reinterpret_cast<const f32x4*>(w3 + (size_t)16 * t2 * H1);
#pragma nounroll
      for (int t1 = 0; t1 < H1T; ++t1) {
        const int tn = t1 + 1 < H1T ? t1 + 1 : t1;
#pragma unroll
        for (int t2 = 0; t2 < JT; ++t2) wn[t2] = *reinterpret_cast<const f32x4*>(w3 + (size_t)16 * t2 * H1 + 16 * tn);
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = wH[64 * t1 + 4 * r];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t2 = 0; t2 < JT; ++t2) dp2[t2] = OB_MFMA(wa[t2][r], h[r], dp2[t2]);
#pragma unroll
        for (int t2 = 0; t2 < JT; ++t2) wa[t2] = wn[t2];
      }
    }
#pragma unroll
    for (int t2 = 0; t2 < JT; ++t2) {
      f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) g = *reinterpret_cast<const f32x4*>(A.dkb + (size_t)p * J + 16 * t2 + 4 * quad);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y, dy;
        ob_gelu_g(dp2[t2][r], y, dy);
        dp2[t2][r] = g[r] * dy;
        wJ[64 * t2 + 4 * r] = dp2[t2][r];
      }
      db3[t2] += dp2[t2];
    }
    __syncthreads();
#ifndef OB_SKIP_W3
    // ---- d W3T[j][h] += sum over the tile's pairs of d pre2[j] h1[h]
#pragma nounroll
    for (int pw = 0; pw < OB_NW; ++pw) {
#pragma unroll
      for (int i = 0; i < NW3; ++i) {
        const int tau = wave + OB_NW * i;
        if (tau < C::NT3) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(rJ + pw * 4 * GSJ + 64 * (tau % JT));
          const f32x4 b = *reinterpret_cast<const f32x4*>(rH + pw * 4 * GSH + 64 * (tau / JT));
#pragma unroll
          for (int r = 0; r < 4; ++r) accW3[i] = OB_MFMA(a[r], b[r], accW3[i]);
        }
      }
    }
#endif
    __syncthreads();                                             // h1 in LDS is dead: the buffer takes d pre1
    // ---- d h1 = W3 d pre2, d pre1 = d h1 * gelu'(pre1)
#pragma unroll
    for (int t = 0; t < H1T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const float* w3n = A.W3 + (size_t)col * J + 4 * quad;
      f32x4 wa[H1T], wn[H1T];
#pragma unroll
      for (int t = 0; t < H1T; ++t) wa[t] = *reinterpret_cast<const f32x4*>(w3n + (size_t)16 * t * J);
#pragma nounroll
      for (int t2 = 0; t2 < JT; ++t2) {
        const int tn = t2 + 1 < JT ? t2 + 1 : t2;
#pragma unroll
        for (int t = 0; t < H1T; ++t) wn[t] = *reinterpret_cast<const f32x4*>(w3n + (size_t)16 * t * J + 16 * tn);
        f32x4 g;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = wJ[64 * t2 + 4 * r];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < H1T; ++t) acc[t] = OB_MFMA(wa[t][r], g[r], acc[t]);
#pragma unroll
        for (int t = 0; t < H1T; ++t) wa[t] = wn[t];
      }
    }
#pragma unroll
    for (int t = 0; t < H1T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (C::STASH) acc[t][r] *= sG[64 * OB_NW * (4 * t + r)];
        else acc[t][r] *= g1[t][r];
        wH[64 * t + 4 * r] = acc[t][r];
      }
    // ---- d features (one 16-feature tile per super-step) -> d inv by the product rule over x_q * mid * x_j
    float dx[4] = {0.f, 0.f, 0.f, 0.f}, dxl = 0.f;
    const float* w1p = A.W1P + (size_t)col * H1 + 4 * quad;
    f32x4 wf0[H1T], wf1[H1T], wf2[H1T];                         // ring of three, two tiles ahead (as above)
    auto ldf = [&](f32x4 (&w)[H1T], int S) {
      const int Sc = S <= NS ? S : NS;
#pragma unroll
      for (int t = 0; t < H1T; ++t) w[t] = *reinterpret_cast<const f32x4*>(w1p + (size_t)16 * Sc * H1 + 16 * t);
    };
    auto ftile = [&](const f32x4 (&w)[H1T], int S) {
      f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
      for (int t = 0; t < H1T; ++t) {
        if (t & 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) d1 = OB_MFMA(w[t][r], acc[t][r], d1);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) d0 = OB_MFMA(w[t][r], acc[t][r], d0);
        }
      }
      const f32x4 d = d0 + d1;                                   // d[r]: feature (q = quad, mid_S, j = r); tile NS: x_r on quad 0
      // branch-free (see ob_mid_idx): tile NS (degree 1) keeps only quad 0's d[r] -> d x_r, the others use the product rule
      const bool last = S == NS;
      const float keep = last ? 0.f : 1.f, first = (last && quad == 0) ? 1.f : 0.f;
      int ia, ib;
      ob_mid_idx<I>(last ? 0 : S, ia, ib);
      const float pa = ob_pick1(x, ia), pb = ob_pick1(x, ib), m = pa * pb * keep;
      const float ts = d[0] * x[0] + d[1] * x[1] + d[2] * x[2] + d[3] * x[3], lm = xl * m, dm = ts * xl * keep;
#pragma unroll
      for (int r = 0; r < 4; ++r) dx[r] += d[r] * (lm + first);
      dxl += ts * m;
#pragma unroll
      for (int c = 0; c < 4; ++c) dx[c] += dm * ((ia == c + 1 ? pb : 0.f) + (ib == c + 1 ? pa : 0.f));
    };
#ifndef OB_SKIP_F
    ldf(wf0, 0);
    ldf(wf1, 1);
    {
      int S = 0;
#pragma nounroll
      for (; S + 3 <= NS + 1; S += 3) {
        ldf(wf2, S + 2); ftile(wf0, S);
        ldf(wf0, S + 3); ftile(wf1, S + 1);
        ldf(wf1, S + 4); ftile(wf2, S + 2);
      }
      if (S <= NS) ftile(wf0, S);
      if (S + 1 <= NS) ftile(wf1, S + 1);
    }
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = dx[c] + (quad == c ? dxl : 0.f);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (c < I && quad == 0 && ok) A.dinv[p * I + c] = v;
    }
    __syncthreads();
#ifndef OB_SKIP_W1
    // ---- d W1T[h][packed feature] += sum over the tile's pairs of d pre1[h] * feature
#pragma nounroll
    for (int pw = 0; pw < OB_NW; ++pw) {
      f32x4 xq[4], lt, one;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xq[r] = *reinterpret_cast<const f32x4*>(sX + (16 * pw + 4 * quad + r) * 4);
        lt[r] = ob_pick4(xq[r], col >> 2) * ob_pick4(xq[r], col & 3);
        one[r] = col < 4 ? ob_pick4(xq[r], col) : (col == 4 ? 1.f : 0.f);
      }
      if constexpr (C::HPW > 0) {                                // wave w owns h tiles w, w + 4, .. for EVERY super-step: S is static
        f32x4 a[C::HPW];
#pragma unroll
        for (int k = 0; k < C::HPW; ++k) a[k] = *reinterpret_cast<const f32x4*>(rH + pw * 4 * GSH + 64 * (wave + OB_NW * k));
#pragma unroll
        for (int S = 0; S <= NS; ++S) {
          f32x4 b = one;
          if (S < NS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = lt[r] * ob_mid4<I>(xq[r], S);
          }
#pragma unroll
          for (int k = 0; k < C::HPW; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) accW1[S * C::HPW + k] = OB_MFMA(a[k][r], b[r], accW1[S * C::HPW + k]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NW1; ++i) {
          const int tau = wave + OB_NW * i;
          if (tau < C::NT1) {
            const int ht = tau % H1T, S = tau / H1T;
            const f32x4 a = *reinterpret_cast<const f32x4*>(rH + pw * 4 * GSH + 64 * ht);
            f32x4 b = one;
            if (S < NS) {
#pragma unroll
              for (int r = 0; r < 4; ++r) b[r] = lt[r] * ob_mid4<I>(xq[r], S);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) accW1[i] = OB_MFMA(a[r], b[r], accW1[i]);
          }
        }
      }
    }
#endif
    __syncthreads();                                             // before the next tile rewrites the buffers
  }

  // ---- this workgroup's partial sums
  float* part = A.part + (size_t)blockIdx.x * C::PART;
#pragma unroll
  for (int i = 0; i < NW1; ++i) {
    const int tau = C::HPW > 0 ? (i / (C::HPW > 0 ? C::HPW : 1)) * H1T + wave + OB_NW * (i % (C::HPW > 0 ? C::HPW : 1))
                               : wave + OB_NW * i;                      // = S H1T + ht
    if (tau < C::NT1) {
      const int ht = tau % H1T, S = tau / H1T;
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(size_t)(16 * ht + 4 * quad + r) * FP + 16 * S + col] = accW1[i][r];
    }
  }
  float* part3 = part + H1 * FP;
#pragma unroll
  for (int i = 0; i < NW3; ++i) {
    const int tau = wave + OB_NW * i;
    if (tau < C::NT3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part3[(size_t)(16 * (tau % JT) + 4 * quad + r) * H1 + 16 * (tau / JT) + col] = accW3[i][r];
    }
  }
  float* pb3 = part3 + J * H1 + wave * J;
#pragma unroll
  for (int t2 = 0; t2 < JT; ++t2) {
    f32x4 v = db3[t2];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], o, 64);
    }
    if (col == 0) *reinterpret_cast<f32x4*>(pb3 + 16 * t2 + 4 * quad) = v;
  }
}

// Sums the workgroup partials in a fixed order and un-packs: d W1 (F, H1), d b1 (H1), d W3 (H1, J), d b3 (J).
struct ObReduceArgs { const float* part; int nwg, I, H1, J, FP, PART; float* dW1; float* db1; float* dW3; float* db3; };
__global__ __launch_bounds__(256) void enf_ode_basis_reduce_kernel(ObReduceArgs A) {
  __shared__ float red[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  const int n1 = A.H1 * A.FP, n3 = A.J * A.H1, n = n1 + n3 + A.J;
  float s = 0.f;
  if (e < n1 + n3) {
    for (int w = grp; w < A.nwg; w += 8) s += A.part[(size_t)w * A.PART + e];
  } else if (e < n) {                                           // d b3: one row per wave of a workgroup
    const int j = e - n1 - n3;
    for (int w = grp; w < A.nwg; w += 8)
      for (int v = 0; v < OB_NW; ++v) s += A.part[(size_t)w * A.PART + n1 + n3 + v * A.J + j];
  }
  red[grp][threadIdx.x & 31] = s;
  __syncthreads();
  if (grp == 0 && e < n) {
    float t = 0.f;
    for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x & 31];
    if (e < n1) {
      const int h = e / A.FP, row = ob_orig_row(A.I, e % A.FP);
      if (row >= 0) A.dW1[(size_t)row * A.H1 + h] = t;
      else if (row == -2) A.db1[h] = t;
    } else if (e < n1 + n3) {
      const int k = e - n1, j = k / A.H1, h = k % A.H1;
      A.dW3[(size_t)h * A.J + j] = t;
    } else {
      A.db3[e - n1 - n3] = t;
    }
  }
}

// --------------------------------------------------------------------------------------------------------------- host
static int ob_check(int64_t P, int I, int degree, int H1, int J, bool backward) {
  if (P <= 0 || P > (1ll << 40)) return ENF_EINVAL;
  if (I < 1 || I > 4 || degree != 3) return ENF_EUNSUPPORTED;
  if (H1 != 32 && H1 != 64 && H1 != 128 && !(H1 == 256 && !backward)) return ENF_EUNSUPPORTED;
  if (J != 32 && J != 64 && J != 128) return ENF_EUNSUPPORTED;
  return ENF_OK;
}
static int ob_bwd_wgs(int64_t P) {
  const long tiles = (long)((P + 16 * OB_NW - 1) / (16 * OB_NW));
  return (int)(tiles < 256 ? tiles : 256);
}

extern "C" int enf_ode_basis_supported(int I, int degree, int H1, int J, int backward) {
  return ob_check(1, I, degree, H1, J, backward != 0) == ENF_OK ? 1 : 0;
}

extern "C" size_t enf_ode_basis_scratch_bytes(int64_t P, int I, int H1, int J, int backward) {
  if (ob_check(P, I, 3, H1, J, backward != 0) != ENF_OK) return 0;
  const size_t FP = 16 * (size_t)(1 + I + I * I + 1);
  size_t words = 2 * FP * H1 + (size_t)J * H1;
  if (backward) words += (size_t)ob_bwd_wgs(P) * (H1 * FP + (size_t)J * H1 + OB_NW * (size_t)J);
  return 4 * words;
}

static void ob_pack(int I, int H1, int J, const float* W1, const float* W3, float* scratch, hipStream_t st) {
  const int FP = 16 * (1 + I + I * I + 1);
  ObPackArgs K{W1, W3, scratch, scratch + (size_t)FP * H1, scratch + 2 * (size_t)FP * H1, I, H1, J, FP};
  hipLaunchKernelGGL(enf_ode_basis_pack_kernel, dim3((FP * H1 + J * H1 + 255) / 256), dim3(256), 0, st, K);
}

#ifndef OB_PT_OVERRIDE
#define OB_PT_OVERRIDE 0   // experiment: pair tiles per wave of the forward kernel (measured at 128 / 64 / I = 4: 1 -> 125 us, 2 -> 84 us, 4 -> 80 us at one wave per SIMD)
#endif
#define OB_SHAPES(X, I_) X(I_, 2, 2) X(I_, 4, 2) X(I_, 4, 4) X(I_, 8, 4) X(I_, 8, 8) X(I_, 2, 4) X(I_, 4, 8) X(I_, 2, 8) X(I_, 8, 2)
#ifdef OB_ONLY_BENCH_SHAPE   // build-time aid: one instantiation
#define OB_ALL(X) X(4, 8, 4)
#else
#define OB_ALL(X) OB_SHAPES(X, 1) OB_SHAPES(X, 2) OB_SHAPES(X, 3) OB_SHAPES(X, 4)
#endif

extern "C" int enf_ode_basis_forward(int64_t P, int I, int degree, int H1, int J, const float* inv, const float* W1,
                                     const float* b1, const float* W3, const float* b3, float* kb, void* scratch,
                                     size_t scratch_bytes, void* stream) {
  int rc = ob_check(P, I, degree, H1, J, false);
  if (rc) return rc;
  if (!inv || !W1 || !b1 || !W3 || !b3 || !kb || !scratch) return ENF_EINVAL;
  if (scratch_bytes < enf_ode_basis_scratch_bytes(P, I, H1, J, 0) || ((uintptr_t)scratch & 15)) return ENF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* sc = (float*)scratch;
  const int FP = 16 * (1 + I + I * I + 1);
  ob_pack(I, H1, J, W1, W3, sc, st);
  ObArgs A{inv, (long)P, sc, sc + (size_t)FP * H1, b1, sc + 2 * (size_t)FP * H1, W3, b3, kb, nullptr, nullptr, nullptr, 0};
  const int H1T = H1 / 16, JT = J / 16;
  bool done = false;
#define OB_FWD(I_, H_, J_)                                                                                              \
  if (!done && I == I_ && H1T == H_ && JT == J_) {                                                                      \
    constexpr int PT = OB_PT_OVERRIDE > 0 ? OB_PT_OVERRIDE : (H_ <= 8 ? 2 : 1);                                         \
    hipLaunchKernelGGL((enf_ode_basis_fwd_kernel<I_, H_, J_, PT>), dim3((unsigned)((P + 64 * PT - 1) / (64 * PT))),      \
                       dim3(256), 0, st, A);                                                                            \
    done = true;                                                                                                        \
  }
  OB_ALL(OB_FWD)
#ifndef OB_ONLY_BENCH_SHAPE
  OB_FWD(4, 16, 8) OB_FWD(4, 16, 4) OB_FWD(3, 16, 8) OB_FWD(3, 16, 4)
#endif
#undef OB_FWD
  if (!done) return ENF_EUNSUPPORTED;
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}

template <int I, int H1T, int JT> static int ob_launch_bwd(ObArgs A, int nwg, hipStream_t st) {
  using C = ObBwd<I, H1T, JT>;
  static EnfAttrBits attr{0};
  if (!enf_lds_attr((const void*)enf_ode_basis_bwd_kernel<I, H1T, JT>, C::LDS_BYTES, attr)) return ENF_ELAUNCH;
  hipLaunchKernelGGL((enf_ode_basis_bwd_kernel<I, H1T, JT>), dim3(nwg), dim3(64 * OB_NW), C::LDS_BYTES, st, A);
  return ENF_OK;
}

extern "C" int enf_ode_basis_backward(int64_t P, int I, int degree, int H1, int J, const float* inv, const float* W1,
                                      const float* b1, const float* W3, const float* b3, const float* dkb, float* dinv,
                                      float* dW1, float* db1, float* dW3, float* db3, void* scratch, size_t scratch_bytes,
                                      void* stream) {
  int rc = ob_check(P, I, degree, H1, J, true);
  if (rc) return rc;
  if (!inv || !W1 || !b1 || !W3 || !b3 || !dkb || !dinv || !dW1 || !db1 || !dW3 || !db3 || !scratch) return ENF_EINVAL;
  if (scratch_bytes < enf_ode_basis_scratch_bytes(P, I, H1, J, 1) || ((uintptr_t)scratch & 15)) return ENF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* sc = (float*)scratch;
  const int FP = 16 * (1 + I + I * I + 1), nwg = ob_bwd_wgs(P);
  ob_pack(I, H1, J, W1, W3, sc, st);
  float* part = sc + 2 * (size_t)FP * H1 + (size_t)J * H1;
  ObArgs A{inv, (long)P, sc, sc + (size_t)FP * H1, b1, sc + 2 * (size_t)FP * H1, W3, b3, nullptr, dkb, dinv, part,
           (int)((P + 16 * OB_NW - 1) / (16 * OB_NW))};
  const int H1T = H1 / 16, JT = J / 16;
  rc = ENF_EUNSUPPORTED;
  bool done = false;
#define OB_BWD(I_, H_, J_)                                                      \
  if (!done && I == I_ && H1T == H_ && JT == J_) {                              \
    rc = ob_launch_bwd<I_, H_, J_>(A, nwg, st);                                 \
    done = true;                                                                \
  }
  OB_ALL(OB_BWD)
#undef OB_BWD
  if (rc) return rc;
  const int PART = H1 * FP + J * H1 + OB_NW * J, n = H1 * FP + J * H1 + J;
  ObReduceArgs R{part, nwg, I, H1, J, FP, PART, dW1, db1, dW3, db3};
  hipLaunchKernelGGL(enf_ode_basis_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, R);
  return hipGetLastError() == hipSuccess ? ENF_OK : ENF_ELAUNCH;
}
