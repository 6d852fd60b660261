// enf_xtd.hip -- K4: the per-pair weight gradients of the training path (include/enf_hip.h: enf_backward_weights).
//
// Every per-pair weight matrix W of the chain y = x W + b has dW = X^T delta and db = 1^T delta over the PAIR axis
// (P = B Z N rows), X / delta being the layer inputs / pre-activation gradients that K3's STORE instantiation wrote
// (ENF_S_*: 7 + 4H buffers of P x D, bf16 fragments in bf16 mode, fp32 in f32 mode).  This file turns that store into the
// 3 + 3H products and their column sums in ONE launch, plus one small reduction:
//
//   enf_xtd_kernel  grid (products, K-slices) x 256 threads.  A workgroup owns one D x D product over one slice of the pair
//                   axis: 32-pair tiles of X and delta go global -> registers -> LDS (swizzled rows), the MFMA operands come
//                   back TRANSPOSED through ds_read_b64_tr_b16 (the pair axis is the MFMA's K, the stored rows have it as
//                   their row index), accumulators stay in registers for the whole slice (64 VGPRs / lane at D = 128).  The
//                   bias sums ride along as one more operand: an all-ones A tile gives 1^T delta on the matrix pipe.
//                   Bound: HBM -- every stored element is read exactly once (2 * P * D * esize bytes per product), the
//                   MFMA and LDS work of a tile are ~1/4 of its HBM time.
//   enf_xtd_reduce  sums the slices in a fixed order (run-to-run identical results), adds the heads of the mixer product,
//                   undoes the column permutation of the bf16 store (ENF_S_*), and writes / accumulates the fp32 gradients
//                   in the layout of the ENF_P_* tensors.
//
// Why not inside K3: the accumulators of the (3 + 3H) D x D products are 9 * 64 KB = 576 KB in fp32 at D = 128, H = 2 --
// more than the 512 KB vector register file of a CU, with K3 itself needing all of it ("Training path").
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_launch.h"
#include "enf_device.h"

#ifndef ENF_XTD_NT            // read-once store: non-temporal loads
#define ENF_XTD_NT 1
#endif

namespace {

constexpr int XTD_THREADS = 256, XTD_WAVES = 4, XTD_TILE = 32;      // pairs per K-step
constexpr int XTD_MAX_PRODUCTS = 3 + 3 * 4;

struct XtdArgs {
  const void* X[XTD_MAX_PRODUCTS];
  const void* Dl[XTD_MAX_PRODUCTS];
  float* part;               // [slice][product][D + 1 rows][D]: rows 0..D-1 = X^T delta (stored column order), row D = 1^T delta
  long long P;               // rows of every buffer
  long long rows_per_slice;  // multiple of XTD_TILE
  int NP;
};

// byte offset of 16-byte chunk `ch` of row `row` in the LDS image of a 32-row bf16 tile: the XOR spreads the 8 rows a
// transposed read touches per half-wave over all 64 banks (rows are 256 B at D = 128 = all 64 banks, 128 B at D = 64)
template <int D> DEV int xtd_off(int row, int ch) {
  if constexpr (D == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  else return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1));
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
union Frag8 { bf16x8 v; s16x4 h[2]; };

// Transposed read of MFMA operands.  One operand = 16 features (stored columns f0 .. f0 + 15) x 32 pairs of a tile: lane l
// holds feature f0 + l % 16, pairs 8 (l / 16) .. + 7, from two ds_read_b64_tr_b16 (rows 8g .. 8g+3 and 8g+4 .. 8g+7 of the
// lane's 16-lane group g).  The per-lane LDS byte addresses do not depend on the tile, so they are computed once:
template <int D> DEV void xtd_tr_addr(unsigned (&adr)[2], const char* tile, int f0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int c0 = f0 >> 3;                                  // 16-byte chunk of the first of the 16 columns
  const unsigned base = (unsigned)(size_t)(lds_ptr_t)tile;  // (generic pointer into __shared__ memory -> LDS byte address)
  adr[0] = base + xtd_off<D>(8 * g + q, c0 + (p >> 1)) + 8 * (p & 1);
  adr[1] = base + xtd_off<D>(8 * g + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
}
// five operands per statement: ten reads in flight, ONE wait (an asm statement takes at most 30 operands).  The reads and
// their wait sit in one statement so that no compiler-scheduled instruction can touch a destination before it has landed.
DEV void xtd_tr_read5(Frag8 (&f)[5], const unsigned (&a)[5][2]) {
  asm volatile("ds_read_b64_tr_b16 %0, %10\n\tds_read_b64_tr_b16 %1, %11\n\t"
               "ds_read_b64_tr_b16 %2, %12\n\tds_read_b64_tr_b16 %3, %13\n\t"
               "ds_read_b64_tr_b16 %4, %14\n\tds_read_b64_tr_b16 %5, %15\n\t"
               "ds_read_b64_tr_b16 %6, %16\n\tds_read_b64_tr_b16 %7, %17\n\t"
               "ds_read_b64_tr_b16 %8, %18\n\tds_read_b64_tr_b16 %9, %19\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(f[0].h[0]), "=&v"(f[0].h[1]), "=&v"(f[1].h[0]), "=&v"(f[1].h[1]), "=&v"(f[2].h[0]), "=&v"(f[2].h[1]),
                 "=&v"(f[3].h[0]), "=&v"(f[3].h[1]), "=&v"(f[4].h[0]), "=&v"(f[4].h[1])
               : "v"(a[0][0]), "v"(a[0][1]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[2][0]), "v"(a[2][1]), "v"(a[3][0]), "v"(a[3][1]),
                 "v"(a[4][0]), "v"(a[4][1])
               : "memory");
}

template <int D, bool BF16>
__global__ __launch_bounds__(XTD_THREADS, 2) void enf_xtd_kernel(XtdArgs A) {
  constexpr int NT = D / 16;                 // 16-wide feature tiles
  constexpr int TR = NT / XTD_WAVES;         // tile rows (X features) per wave: 2 at D = 128, 1 at D = 64
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int LDF = D + 16;                // fp32 image: padded rows (ds_read_b32 of rows r, r + 1 on distinct banks)
  constexpr int TILE_BYTES = BF16 ? XTD_TILE * D * 2 : XTD_TILE * LDF * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];
  char* tx = smem;
  char* td = smem + TILE_BYTES;
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  // wave-uniform in an SGPR: the `b / TR == wave` tests below must be scalar branches -- MFMA ignores EXEC, so under a
  // lane-masked region it would run (and clobber its accumulator) in the waves that are meant to skip it
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long r0 = (long long)blockIdx.y * A.rows_per_slice;
  const long long r1 = r0 + A.rows_per_slice < A.P ? r0 + A.rows_per_slice : A.P;
  const char* X = reinterpret_cast<const char*>(A.X[j]);
  const char* Dl = reinterpret_cast<const char*>(A.Dl[j]);

  f32x4 acc[TR][NT];
  f32x4 bacc[TR];                            // this wave's share of the bias tiles: delta tiles TR * wave .. + TR - 1
#pragma unroll
  for (int a = 0; a < TR; ++a) {
    bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // global -> registers: 16-byte chunks; a 32-row tile has 32 * D * ES / 16 of them per buffer
  constexpr int CH_ROW = D * ES / 16;                          // chunks per row
  constexpr int NCH = XTD_TILE * CH_ROW / XTD_THREADS;         // chunks per thread and buffer (2 / 1 bf16, 4 / 2 fp32)
  static_assert(XTD_TILE * CH_ROW % XTD_THREADS == 0, "tile chunks");
  // two register sets: the loads of tiles t + 1 and t + 2 are in flight while tile t is on the matrix pipe (at two
  // workgroups per CU that is ~130 KB per CU outstanding, what the HBM latency-bandwidth product asks for)
  f32x4 gxa[NCH], gda[NCH], gxb[NCH], gdb[NCH];
  auto gload = [&](f32x4 (&gx)[NCH], f32x4 (&gd)[NCH], long long row0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = tid + c * XTD_THREADS, row = idx / CH_ROW, ch = idx % CH_ROW;
      const long long gr = row0 + row;
      if (gr < r1) {
#if ENF_XTD_NT
        gx[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + ((size_t)gr * D * ES + 16 * ch)));
        gd[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Dl + ((size_t)gr * D * ES + 16 * ch)));
#else
        gx[c] = *reinterpret_cast<const f32x4*>(X + ((size_t)gr * D * ES + 16 * ch));
        gd[c] = *reinterpret_cast<const f32x4*>(Dl + ((size_t)gr * D * ES + 16 * ch));
#endif
      } else {
        gx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        gd[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto lstore = [&](const f32x4 (&gx)[NCH], const f32x4 (&gd)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = tid + c * XTD_THREADS, row = idx / CH_ROW, ch = idx % CH_ROW;
      int off;
      if constexpr (BF16) off = xtd_off<D>(row, ch);
      else off = (row * LDF + 4 * ch) * 4;
      *reinterpret_cast<f32x4*>(tx + off) = gx[c];
      *reinterpret_cast<f32x4*>(td + off) = gd[c];
    }
  };
  // per-lane LDS addresses of the operand reads (tile-independent)
  constexpr int NG = (TR + NT) / 5;
  static_assert((TR + NT) % 5 == 0, "operand groups of five");
  unsigned adr[NG][5][2];
  if constexpr (BF16) {
#pragma unroll
    for (int o = 0; o < TR + NT; ++o)
      xtd_tr_addr<D>(adr[o / 5][o % 5], o < TR ? tx : td, o < TR ? 16 * (TR * wave + o) : 16 * (o - TR), lane);
  }
  auto compute = [&]() {
    if constexpr (BF16) {
      // operands of a tile in groups of five: [this wave's TR feature tiles of X | the delta tiles], TR + NT = 10 or 5
      Frag8 f[NG][5];
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) xtd_tr_read5(f[gi], adr[gi]);
      Frag8 ones;
      ones.h[0] = s16x4{0x3f80, 0x3f80, 0x3f80, 0x3f80};     // bf16 1.0
      ones.h[1] = ones.h[0];
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const bf16x8 fb = f[(TR + b) / 5][(TR + b) % 5].v;
#pragma unroll
        for (int a = 0; a < TR; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0][a].v, fb, acc[a][b], 0, 0, 0);
        if (b / TR == wave) bacc[b % TR] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, fb, bacc[b % TR], 0, 0, 0);
      }
    } else {
      const float* fx = reinterpret_cast<const float*>(tx);
      const float* fd = reinterpret_cast<const float*>(td);
      const int k = lane >> 4, i = lane & 15;
#pragma unroll 2
      for (int k0 = 0; k0 < XTD_TILE; k0 += 4) {
        float fa[TR];
#pragma unroll
        for (int a = 0; a < TR; ++a) fa[a] = fx[(k0 + k) * LDF + 16 * (TR * wave + a) + i];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const float fb = fd[(k0 + k) * LDF + 16 * b + i];
#pragma unroll
          for (int a = 0; a < TR; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb, acc[a][b], 0, 0, 0);
          if (b / TR == wave) bacc[b % TR] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, fb, bacc[b % TR], 0, 0, 0);
        }
      }
    }
  };

  gload(gxa, gda, r0);                                       // (rows beyond the slice load as zeros)
  gload(gxb, gdb, r0 + XTD_TILE);
  for (long long row0 = r0; row0 < r1; row0 += 2 * XTD_TILE) {
    __syncthreads();                     // the previous tile's operand reads are done
    lstore(gxa, gda);
    __syncthreads();
    gload(gxa, gda, row0 + 2 * XTD_TILE);
    compute();
    if (row0 + XTD_TILE < r1) {
      __syncthreads();
      lstore(gxb, gdb);
      __syncthreads();
      gload(gxb, gdb, row0 + 3 * XTD_TILE);
      compute();
    }
  }

  // partial sums of this (slice, product): C tile element (m = 4 (lane / 16) + i, n = lane % 16)
  float* out = A.part + ((size_t)blockIdx.y * A.NP + j) * (size_t)(D + 1) * D;
  const int g = lane >> 4, n = lane & 15;
#pragma unroll
  for (int a = 0; a < TR; ++a) {
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(size_t)(16 * (TR * wave + a) + 4 * g + i) * D + 16 * b + n] = acc[a][b][i];
    if (g == 0) out[(size_t)D * D + 16 * (TR * wave + a) + n] = bacc[a][0];       // every row of the ones product is 1^T delta
  }
}

// ---- reduction: fixed order over slices (and over the sources of a destination), column un-permutation, fp32 outputs
constexpr int XTD_MAX_DEST = 4 + 2 * 4, XTD_MAX_SRC = 4;
struct XtdReduceArgs {
  const float* part;
  float* w[XTD_MAX_DEST];      // destination matrix block: element (r, c) at w[r * ld + c]
  float* b[XTD_MAX_DEST];      // destination bias block (D values)
  int ld[XTD_MAX_DEST];
  int nsrc[XTD_MAX_DEST];
  int src[XTD_MAX_DEST][XTD_MAX_SRC];
  int ND, NP, KS, accumulate;
};

// stored column of a true feature (ENF_S_*: bf16 rows are permuted inside every 32-block; fp32 rows are not)
template <bool BF16> DEV int xtd_stored_col(int t) {
  if constexpr (!BF16) return t;
  const int b = t >> 5, r = t & 31;
  return r < 16 ? 32 * b + 8 * (r >> 2) + (r & 3) : 32 * b + 8 * ((r - 16) >> 2) + 4 + (r & 3);
}

// 32 output elements x 8 slice groups per workgroup: a thread sums its group's slices (sl = group, group + 8, ...) of every
// source in order, the eight group sums are added in order through LDS -- a fixed summation tree, 8x the loads in flight
template <int D, bool BF16>
__global__ __launch_bounds__(256) void enf_xtd_reduce_kernel(XtdReduceArgs A) {
  __shared__ float red[8][32];
  const int d = blockIdx.y, el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;                     // element of the (D + 1) x D block; row D = the bias
  const bool live = e < (D + 1) * D;
  const int r = live ? e / D : 0, c = live ? e % D : 0;
  const int sr = r < D ? xtd_stored_col<BF16>(r) : D, sc = xtd_stored_col<BF16>(c);
  float s = 0.f;
  if (live)
    for (int k = 0; k < A.nsrc[d]; ++k) {
      const float* p = A.part + ((size_t)A.src[d][k]) * (size_t)(D + 1) * D + (size_t)sr * D + sc;
      for (int sl = grp; sl < A.KS; sl += 8) s += p[(size_t)sl * A.NP * (D + 1) * D];
    }
  red[grp][el] = s;
  __syncthreads();
  if (grp == 0 && live) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][el];
    float* dst = r < D ? A.w[d] + (size_t)r * A.ld[d] + c : A.b[d] + c;
    *dst = A.accumulate ? *dst + t : t;
  }
}

template <int D, bool BF16>
int launch_xtd(const XtdArgs& A, const XtdReduceArgs& R, int KS, hipStream_t st) {
  hipLaunchKernelGGL((enf_xtd_kernel<D, BF16>), dim3(A.NP, KS), dim3(XTD_THREADS), 0, st, A);
  if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
  hipLaunchKernelGGL((enf_xtd_reduce_kernel<D, BF16>), dim3(((D + 1) * D + 31) / 32, R.ND), dim3(256), 0, st, R);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

}  // namespace

// K-slices of a pass over P rows: the launch is ONE round of workgroups, all resident at once (two 4-wave workgroups per CU
// at D = 128, 256 CUs) and all with the same number of tiles, so nobody waits for a straggler; at least 8 tiles per slice
static int xtd_slices(long long P, int NP) {
  const long long tiles = (P + XTD_TILE - 1) / XTD_TILE;
  long long ks = 512 / NP;
  if (ks > tiles / 8) ks = tiles / 8;
  if (ks < 1) ks = 1;
  return (int)ks;
}

size_t enf_xtd_part_bytes(const EnfDims& m, long long P) {
  const int NP = 3 + 3 * m.H;
  return enf_align(sizeof(float) * (size_t)xtd_slices(P, NP) * NP * (m.D + 1) * m.D);
}

// store: the ENF_NUM_STORE(H) device buffers K3 wrote for P rows; dpair: ENF_NUM_PAIR_TENSORS fp32 device pointers in
// ENF_P_* order (the two coefficient entries are not touched); accumulate = add to what dpair holds (later chunks)
int enf_launch_xtd(const EnfDims& m, void* const* store, long long P, float* const* dpair, float* part, int accumulate,
                   hipStream_t st) {
  const int H = m.H, D = m.D, HD = m.HD;
  XtdArgs A;
  XtdReduceArgs R;
  A.P = P; A.part = part; R.part = part; R.accumulate = accumulate;
  int np = 0, nd = 0;
  auto product = [&](int sx, int sd) { A.X[np] = store[sx]; A.Dl[np] = store[sd]; return np++; };
  auto dest = [&](float* w, int ld, float* b, int s0) { R.w[nd] = w; R.ld[nd] = ld; R.b[nd] = b; R.nsrc[nd] = 1; R.src[nd][0] = s0; return nd++; };
  dest(dpair[ENF_P_AQ1], D, dpair[ENF_P_BQ1], product(ENF_S_EQ, ENF_S_DA1));
  dest(dpair[ENF_P_AV1], D, dpair[ENF_P_BV1], product(ENF_S_EV, ENF_S_DA2));
  dest(dpair[ENF_P_AF], D, dpair[ENF_P_BF], product(ENF_S_G1, ENF_S_DA3));
  int mixer = -1;
  for (int h = 0; h < H; ++h) {
    const int s0 = ENF_S_HEAD0 + 4 * h;                   // V, DA5, DG, DB
    dest(dpair[ENF_P_AGB] + h * D, 2 * HD, dpair[ENF_P_BGB] + h * D, product(ENF_S_NH, s0 + 2));
    dest(dpair[ENF_P_AGB] + HD + h * D, 2 * HD, dpair[ENF_P_BGB] + HD + h * D, product(ENF_S_NH, s0 + 3));
    const int pm = product(s0, s0 + 1);
    if (mixer < 0) mixer = dest(dpair[ENF_P_AM], D, dpair[ENF_P_BM], pm);
    else R.src[mixer][R.nsrc[mixer]++] = pm;              // the mixer's gradient sums over the heads, in head order
  }
  A.NP = np; R.NP = np; R.ND = nd;
  const int KS = xtd_slices(P, np);
  const long long tiles = (P + XTD_TILE - 1) / XTD_TILE;
  A.rows_per_slice = ((tiles + KS - 1) / KS) * XTD_TILE;
  R.KS = KS;
  if (D == 128) return m.bf16 ? launch_xtd<128, true>(A, R, KS, st) : launch_xtd<128, false>(A, R, KS, st);
  if (D == 64) return m.bf16 ? launch_xtd<64, true>(A, R, KS, st) : launch_xtd<64, false>(A, R, KS, st);
  return ENF_EUNSUPPORTED;
}
