// enf_train.hip -- enf_backward_all: the gradient of nef.apply w.r.t. EVERY weight tensor and the latents in one library call
// (jax.value_and_grad over params['nef'], experiments/fitting/trainers/pde_trainer.py:255; nonmaml_pde_trainer.py:304-339).
//
//   d out --tail backward (enf_tail.hip, weight-gradient form)--> d ybar, delta, and per query every tail layer's input / delta
//         --K3 (store) + K4 (enf_pair_bwd.hip, enf_xtd.hip)-----> d lt and the ten per-pair tensors' gradients (97 % of the FLOPs)
//         --prologue backward (enf_pack.hip, with operand rows)--> d p, d a, d sigma, and per latent the prologue's inputs / deltas
//         --X^T delta products + column sums (this file)---------> gradients of the FOLDED tail / prologue matrices
//         --fold backward (this file)-----------------------------> the 46 Flax-named tensors (exact chain rule through
//                                                                    enf_pack_weights' folds, enf_layout.h)
// Everything is a kernel of this library on the caller's stream: no library GEMM, no host-framework op.  The products here are
// small (rows = B N queries or B Z latents, at most 256 x 256 outputs): fp32 on the matrix pipe (v_mfma_f32_16x16x4_f32), split
// over row slices, slices summed in a fixed order -- same inputs, same bits.
#include <hip/hip_runtime.h>
#include "enf_layout.h"
#include "enf_launch.h"

int enf_side_join_pending(hipStream_t st, const void* workspace);      // enf_api.hip
extern "C" {
int enf_launch_prologue(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*, float*,
                        float*, float*, hipStream_t);
int enf_launch_prologue_bwd_wg(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*,
                               const float*, const float*, float*, float*, float*, float*, hipStream_t);
int enf_launch_pair_bwd(const EnfDims&, const EnfLayout&, const char*, const float*, long long, const float*, const float*,
                        const float*, const float*, float*, void* const*, const char*, const float*, float*, hipStream_t);
int enf_launch_tail_wg(const EnfDims&, const EnfLayout&, const char*, const float*, float*, const float*, float*, float*, float*,
                       float*, int, int, hipStream_t);
}
size_t enf_xtd_part_bytes(const EnfDims& m, long long P);
int enf_launch_xtd(const EnfDims& m, void* const* store, long long P, float* const* dpair, float* part, int accumulate, hipStream_t st);

typedef float tf4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ X^T Y over long row axes
// part[s][k][n] = sum_{r in slice s} X[r][k] Y[r][n]     (K x N tile of 64 x 64 per workgroup, 16 x 64 per wave)
// v_mfma_f32_16x16x4_f32: A[i][kk] = X[r0 + kk][k0 + i] (lane: i = lane & 15, kk = lane >> 4), B[kk][j] = Y[r0 + kk][n0 + j].
constexpr int XT_ROWS = 32;       // rows per unrolled step (8 MFMA k-steps of 4 rows: 8 + 32 loads in flight per lane)
constexpr int XT_U = XT_ROWS / 4;
__global__ __launch_bounds__(256) void enf_rows_xty_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ Y,
                                                          long long ldy, long long R, int K, int N, long long rows_per_slice,
                                                          float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 64 + wave * 16, n0 = blockIdx.y * 64;
  const long long r_lo = (long long)blockIdx.z * rows_per_slice;
  const long long r_hi = r_lo + rows_per_slice < R ? r_lo + rows_per_slice : R;
  tf4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = tf4{0.f, 0.f, 0.f, 0.f};
  const bool kok = k0 + i < K;
  bool nok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) nok[t] = n0 + 16 * t + i < N;
  for (long long r0 = r_lo; r0 < r_hi; r0 += XT_ROWS) {
    float xa[XT_U], yb[XT_U][4];
#pragma unroll
    for (int u = 0; u < XT_U; ++u) {
      const long long r = r0 + 4 * u + q;
      const bool rok = r < r_hi;
      xa[u] = rok && kok ? X[r * ldx + k0 + i] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) yb[u][t] = rok && nok[t] ? Y[r * ldy + n0 + 16 * t + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < XT_U; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u], yb[u][t], acc[t], 0, 0, 0);
  }
  // C[ii][j]: lane holds rows ii = 4 q + e, column j = i
  float* o = part + (size_t)blockIdx.z * K * N;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + 4 * q + e, n = n0 + 16 * t + i;
      if (k < K && n < N) o[(size_t)k * N + n] = acc[t][e];
    }
}
// out[k][n] = (acc ? out : 0) + alpha * sum_s part[s][k][n]   (fixed order)
__global__ void enf_slices_sum_kernel(const float* __restrict__ part, int S, long long KN, float* __restrict__ out, float alpha, int acc) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= KN) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += part[(size_t)z * KN + e];
  out[e] = acc ? out[e] + alpha * s : alpha * s;
}
// part[s][n] = sum_{r in slice s} Y[r][n] (* X[r][n] if X): 64 columns x 4 row groups per workgroup, slices summed afterwards in a
// fixed order (enf_slices_sum_kernel)
__global__ __launch_bounds__(256) void enf_rows_colsum_kernel(const float* __restrict__ Y, long long ldy, const float* __restrict__ X,
                                                             long long ldx, long long R, int N, long long rows_per_slice,
                                                             float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6, n = blockIdx.x * 64 + c;
  const long long r_lo = (long long)blockIdx.y * rows_per_slice;
  const long long r_hi = r_lo + rows_per_slice < R ? r_lo + rows_per_slice : R;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    long long r = r_lo + g;
    for (; r + 12 < r_hi; r += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rr = r + 4 * u;
        s[u] += X ? Y[rr * ldy + n] * X[rr * ldx + n] : Y[rr * ldy + n];
      }
    }
    for (; r < r_hi; r += 4) s[0] += X ? Y[r * ldy + n] * X[r * ldx + n] : Y[r * ldy + n];
  }
  red[g][c] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (g == 0 && n < N) part[(size_t)blockIdx.y * N + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ------------------------------------------------------------------------------------------------ small dense helpers (weight space)
// C (M x N) = (acc ? C : 0) + alpha * op(A) op(B),   op(A) = ta ? A^T : A  (A stored (ta ? K x M : M x K) with leading dimension lda)
__global__ __launch_bounds__(256) void enf_small_gemm_kernel(float* __restrict__ C, int ldc, const float* __restrict__ A, int lda, int ta,
                                                            const float* __restrict__ B, int ldb, int tb, int M, int N, int K, float alpha, int acc) {
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  float s = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    {   // sa[ty][tx] = opA[m0 + ty][k0 + tx]
      const int mm = blockIdx.y * 16 + ty, kk = k0 + tx;
      sa[ty][tx] = (mm < M && kk < K) ? (ta ? A[(size_t)kk * lda + mm] : A[(size_t)mm * lda + kk]) : 0.f;
      const int kb = k0 + ty, nn = blockIdx.x * 16 + tx;
      sb[ty][tx] = (kb < K && nn < N) ? (tb ? B[(size_t)nn * ldb + kb] : B[(size_t)kb * ldb + nn]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) s = fmaf(sa[ty][k], sb[k][tx], s);
    __syncthreads();
  }
  if (m < M && n < N) C[(size_t)m * ldc + n] = acc ? C[(size_t)m * ldc + n] + alpha * s : alpha * s;
}
// out[i] = (acc ? out[i] : 0) + sum_j A[i][j] B[i][j]      (rows x cols; one wave per row)
__global__ __launch_bounds__(256) void enf_rowdot_kernel(float* out, const float* A, int lda, const float* B, int ldb, int rows, int cols, int acc) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) s = fmaf(A[(size_t)row * lda + j], B[(size_t)row * ldb + j], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[row] = acc ? out[row] + s : s;
}
// C[i][j] = (acc ? C : 0) + g[i] * A[i][j] (g may be NULL = 1) + (u ? u[i] * v[j] : 0)
__global__ void enf_scale_outer_kernel(float* C, int ldc, const float* A, int lda, const float* g, const float* u, const float* v, int rows,
                                       int cols, int acc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= cols || i >= rows) return;
  float s = acc ? C[(size_t)i * ldc + j] : 0.f;
  if (A) s += (g ? g[i] : 1.f) * A[(size_t)i * lda + j];
  if (u) s += u[i] * v[j];
  C[(size_t)i * ldc + j] = s;
}
// dst[i] = alpha * src[i] (+ dst[i] if acc)
__global__ void enf_axpy_kernel(float* dst, const float* src, int n, float alpha, int acc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = acc ? dst[i] + alpha * src[i] : alpha * src[i];
}

namespace {
struct Ctx {
  hipStream_t st;
  float* part;          // X^T Y slice partials
  size_t part_floats;
  int rc = 0;
  bool ok() const { return rc == 0; }
  void chk() { if (hipGetLastError() != hipSuccess) rc = ENF_ELAUNCH; }
  // out (K x N, row-major, leading dimension N) = alpha * X^T Y over R rows
  void xty(const float* X, long long ldx, const float* Y, long long ldy, long long R, int K, int N, float* out, float alpha = 1.f, int acc = 0) {
    if (rc) return;
    const size_t KN = (size_t)K * N;
    long long S = (R + 511) / 512;                       // slices of ~512 rows; bounded by the partial buffer and by 64
    if (S > 64) S = 64;
    while (S > 1 && (size_t)S * KN > part_floats) --S;
    if ((size_t)S * KN > part_floats) { rc = ENF_EWORKSPACE; return; }
    long long rps = ((R + S - 1) / S + XT_ROWS - 1) / XT_ROWS * XT_ROWS;
    S = (R + rps - 1) / rps;
    hipLaunchKernelGGL(enf_rows_xty_kernel, dim3((K + 63) / 64, (N + 63) / 64, (unsigned)S), dim3(256), 0, st, X, ldx, Y, ldy, R, K, N, rps, part);
    hipLaunchKernelGGL(enf_slices_sum_kernel, dim3((unsigned)((KN + 255) / 256)), dim3(256), 0, st, part, (int)S, (long long)KN, out, alpha, acc);
    chk();
  }
  void colsum(const float* Y, long long ldy, long long R, int N, float* out, const float* X = nullptr, long long ldx = 0, float alpha = 1.f, int acc = 0) {
    if (rc) return;
    long long S = (R + 255) / 256;
    if (S > 64) S = 64;
    const long long rps = (R + S - 1) / S;
    S = (R + rps - 1) / rps;
    hipLaunchKernelGGL(enf_rows_colsum_kernel, dim3((N + 63) / 64, (unsigned)S), dim3(256), 0, st, Y, ldy, X, ldx, R, N, rps, part);
    hipLaunchKernelGGL(enf_slices_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, st, part, (int)S, (long long)N, out, alpha, acc);
    chk();
  }
  void gemm(float* C, int ldc, const float* A, int lda, int ta, const float* B, int ldb, int tb, int M, int N, int K, float alpha = 1.f, int acc = 0) {
    if (rc) return;
    hipLaunchKernelGGL(enf_small_gemm_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, st, C, ldc, A, lda, ta, B, ldb, tb, M, N, K, alpha, acc);
    chk();
  }
  void rowdot(float* out, const float* A, int lda, const float* B, int ldb, int rows, int cols, int acc = 0) {
    if (rc) return;
    hipLaunchKernelGGL(enf_rowdot_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, out, A, lda, B, ldb, rows, cols, acc);
    chk();
  }
  // C = g (.) A (row scale) + u (x) v
  void scale_outer(float* C, int ldc, const float* A, int lda, const float* g, const float* u, const float* v, int rows, int cols, int acc = 0) {
    if (rc) return;
    hipLaunchKernelGGL(enf_scale_outer_kernel, dim3((cols + 127) / 128, rows), dim3(128), 0, st, C, ldc, A, lda, g, u, v, rows, cols, acc);
    chk();
  }
  void axpy(float* dst, const float* src, int n, float alpha = 1.f, int acc = 0) {
    if (rc) return;
    hipLaunchKernelGGL(enf_axpy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, n, alpha, acc);
    chk();
  }
};

// scratch carving (bytes, 256-aligned pieces)
struct TrainScratch {
  size_t pair;      // K3 store buffers + K4 partials of one chunk (enf_backward_weights' scratch)
  size_t tdel;      // B N x (2 HD + 2 D): tail deltas
  size_t pg;        // B Z x (2 HD + 4 D): prologue operand rows
  size_t dpair;     // the ten per-pair gradients, ENF_P_* order, contiguous
  size_t fold;      // folded-matrix gradients and temporaries
  size_t part;      // X^T Y slice partials
  size_t total;
  size_t part_floats, pair_bytes;
};
inline size_t pair_grad_floats(const EnfDims& m) {
  const size_t D = m.D, HD = m.HD;
  return 4 * D * D + 4 * D + D * 2 * HD + 2 * HD;        // AQ1 BQ1 AV1 BV1 AF BF AGB BGB AM BM
}
inline size_t fold_floats(const EnfDims& m) {
  const size_t D = m.D, HD = m.HD, H = m.H;
  // dWB HDxHD | dbB HD | dwf1 HDxHD | dbF1 HD | T HDxHD | dT HDxHD | dMU H DxD | dcvec HD | dmxw DxD | dmxb D
  return 4 * HD * HD + 3 * HD + H * D * D + D * D + D + 256;
}
size_t bw_store_bytes_(const EnfDims& m, int cb) { return enf_align((size_t)cb * m.Z * m.N * m.D * (m.bf16 ? 2 : 4)); }
size_t bw_pair_bytes_(const EnfDims& m, int cb) {
  return (size_t)ENF_NUM_STORE(m.H) * bw_store_bytes_(m, cb) + enf_align(enf_xtd_part_bytes(m, (long long)cb * m.Z * m.N));
}
TrainScratch train_scratch(const EnfDims& m, int cb) {
  TrainScratch s;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = enf_align(o + bytes); return r; };
  s.pair_bytes = bw_pair_bytes_(m, cb);
  s.pair = take(s.pair_bytes);
  s.tdel = take(sizeof(float) * (size_t)m.B * m.N * (2 * m.HD + 2 * m.D));
  s.pg = take(sizeof(float) * (size_t)m.B * m.Z * (2 * m.HD + 4 * m.D));
  s.dpair = take(sizeof(float) * pair_grad_floats(m));
  s.fold = take(sizeof(float) * fold_floats(m));
  s.part_floats = (size_t)64 * m.HD * m.HD;
  s.part = take(sizeof(float) * s.part_floats);
  s.total = o;
  return s;
}
}  // namespace

extern "C" size_t enf_backward_all_scratch_bytes(const EnfDesc* d, int chunk_signals) {
  if (enf_check_desc(d) != ENF_OK || chunk_signals < 1 || chunk_signals > d->B) return 0;
  return train_scratch(enf_dims(d), chunk_signals).total;
}

extern "C" int enf_backward_all(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a, const float* sigma,
                                const float* const* T, const void* packed, const float* ybar, const float* lse, const float* dout,
                                float* dp, float* da, float* dsigma, float* const* dW, float* dx, void* workspace,
                                size_t workspace_bytes, void* scratch, size_t scratch_bytes, unsigned flags, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !p || !a || !T || !packed || !ybar || !lse || !dout || !dp || !da || !dsigma || !dW || !workspace || !scratch) return ENF_EINVAL;
  if (d->use_window && !sigma) return ENF_EINVAL;
  for (int i = 0; i < ENF_NUM_TENSORS; ++i) {
    if (!T[i]) return ENF_EINVAL;
    if (!dW[i] && i != ENF_W_RQ_COEF && i != ENF_W_RV_COEF) return ENF_EINVAL;
  }
  EnfDims m = enf_dims(d);
  if (m.OB != 1) return ENF_EUNSUPPORTED;
  const EnfLayout L = enf_layout(m);
  const EnfWorkspace W = enf_workspace(m);
  if (workspace_bytes < W.total) return ENF_EWORKSPACE;
  // the largest chunk of signals whose activation store fits (with relu masks: whole groups of mask_signals)
  const int step = m.mask_mode == ENF_MASK_READ && m.mask_B < m.B ? m.mask_B : 1;
  int cb = m.B;
  while (cb > step && train_scratch(m, cb).total > scratch_bytes) cb = (cb - 1) / step * step;
  if (cb < 1) return ENF_EWORKSPACE;
  const TrainScratch S = train_scratch(m, cb);
  if (S.total > scratch_bytes) return ENF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  char* sc = (char*)scratch;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  auto G = [&](size_t off) { return reinterpret_cast<float*>(sc + off); };
  const char* blob = (const char*)packed;
  auto Bf = [&](size_t off) { return reinterpret_cast<const float*>(blob + off); };
  const int D = m.D, H = m.H, HD = m.HD, C = m.C, O = m.O;
  const long long BN = (long long)m.B * m.N, BZ = (long long)m.B * m.Z;
  const int stride = enf_lt_stride(H, D);

  // ---- 1. latent table (unless the matching forward left it), tail backward in its weight-gradient form
  if ((rc = enf_side_join_pending(st, workspace))) return rc;      // (side-stream work an earlier inner-loop forward left on this workspace)
  if (!(flags & ENF_BWD_REUSE_PROLOGUE) && (rc = enf_launch_prologue(m, L, blob, p, a, sigma, F(W.lt), F(W.an), F(W.kv), st))) return rc;
  const bool treuse = (flags & ENF_BWD_REUSE_TAIL) && (flags & ENF_BWD_REUSE_PROLOGUE);
  float* tdel = G(S.tdel);
  if ((rc = enf_launch_tail_wg(m, L, blob, ybar, nullptr, dout, F(W.dybar), F(W.delta), F(W.tail_act), tdel, 1, treuse ? 1 : 0, st))) return rc;

  // ---- 2. the per-pair chain: K3 with the activation store, K4 (chunked over signals like enf_backward_weights)
  if (hipMemsetAsync(F(W.dlt), 0, sizeof(float) * (size_t)BZ * stride, st) != hipSuccess) return ENF_ELAUNCH;
  float* dpair[ENF_NUM_PAIR_TENSORS];
  {
    float* q = G(S.dpair);
    const size_t sz[10] = {(size_t)D * D, (size_t)D, (size_t)D * D, (size_t)D, (size_t)D * D, (size_t)D, (size_t)D * 2 * HD, (size_t)2 * HD,
                           (size_t)D * D, (size_t)D};
    for (int i = 0; i < 10; ++i) { dpair[i] = q; q += sz[i]; }
    dpair[ENF_P_COEFQ] = dpair[ENF_P_COEFV] = nullptr;
  }
  {
    void* store[ENF_NUM_STORE(4)];
    const size_t sb = bw_store_bytes_(m, cb);
    for (int i = 0; i < ENF_NUM_STORE(H); ++i) store[i] = sc + S.pair + (size_t)i * sb;
    float* part = reinterpret_cast<float*>(sc + S.pair + (size_t)ENF_NUM_STORE(H) * sb);
    for (int b0 = 0; b0 < m.B; b0 += cb) {
      const int nb = b0 + cb <= m.B ? cb : m.B - b0;
      EnfDims mc = m;
      mc.B = nb; mc.mask_b0 = b0;
      const size_t qo = (size_t)b0 * m.N;
      if ((rc = enf_launch_pair_bwd(mc, L, blob, x + (size_t)b0 * x_bstride, x_bstride, F(W.lt) + (size_t)b0 * m.Z * stride,
                                    lse + qo * H, F(W.dybar) + qo * HD, F(W.delta) + qo * H, F(W.dlt) + (size_t)b0 * m.Z * stride, store,
                                    nullptr, nullptr, dx ? dx + qo * m.dx : nullptr, st)))
        return rc;
      if ((rc = enf_launch_xtd(mc, store, (long long)nb * m.Z * m.N, dpair, part, b0 > 0, st))) return rc;
    }
  }

  // ---- 3. prologue backward: d p, d a, d sigma + the operand rows of its weight gradients
  float* pg = G(S.pg);
  const int PGW = 2 * HD + 4 * D;
  if ((rc = enf_launch_prologue_bwd_wg(m, L, blob, p, sigma, F(W.an), F(W.kv), F(W.dlt), dp, da, dsigma, pg, st))) return rc;

  Ctx X;
  X.st = st; X.part = G(S.part); X.part_floats = S.part_floats;
  float* fo = G(S.fold);
  float* dWB = fo; fo += (size_t)HD * HD;
  float* dbB = fo; fo += HD;
  float* dwf1 = fo; fo += (size_t)HD * HD;
  float* dbF1 = fo; fo += HD;
  float* Tm = fo; fo += (size_t)HD * HD;        // AO_W FF_W0
  float* dT = fo; fo += (size_t)HD * HD;
  float* dMU = fo; fo += (size_t)H * D * D;
  float* dcv = fo; fo += HD;
  float* dmxw = fo; fo += (size_t)D * D;
  float* dmxb = fo; fo += D;

  // ---- 4. tail: dW = X^T delta per (folded) layer, biases = column sums.  act now holds n^ | gelu(a_F1) | gelu(a_O0) | gelu(a_O2)
  const float* act = F(W.tail_act);
  const int ACT = 2 * HD + 2 * D + 2, TD = 2 * HD + 2 * D;
  X.xty(ybar, HD, tdel, TD, BN, HD, HD, dWB);                     X.colsum(tdel, TD, BN, HD, dbB);
  X.xty(act, ACT, tdel + HD, TD, BN, HD, HD, dwf1);               X.colsum(tdel + HD, TD, BN, HD, dbF1);
  X.xty(act + HD, ACT, tdel + 2 * HD, TD, BN, HD, D, dW[ENF_W_O0_W]);      X.colsum(tdel + 2 * HD, TD, BN, D, dW[ENF_W_O0_B]);
  X.xty(act + 2 * HD, ACT, tdel + 2 * HD + D, TD, BN, D, D, dW[ENF_W_O2_W]);   X.colsum(tdel + 2 * HD + D, TD, BN, D, dW[ENF_W_O2_B]);
  X.xty(act + 2 * HD + D, ACT, dout, O, BN, D, O, dW[ENF_W_O4_W]);         X.colsum(dout, O, BN, O, dW[ENF_W_O4_B]);

  // ---- 5. prologue: rows = latents.  pg = [d k | d an | d s | an | d an * xn | d c k];  kv = [k | v0];  d lt = [d u | d v0 | ..]
  const float* dlt = F(W.dlt);
  const float* kv = F(W.kv);
  X.xty(pg + HD + 2 * D, PGW, pg, PGW, BZ, D, HD, dW[ENF_W_K_W]);          X.colsum(pg, PGW, BZ, HD, dW[ENF_W_K_B]);
  X.xty(pg + HD + 2 * D, PGW, dlt + enf_lt_off_v0(H, D), stride, BZ, D, HD, dW[ENF_W_V_W]);
  X.colsum(dlt + enf_lt_off_v0(H, D), stride, BZ, HD, dW[ENF_W_V_B]);
  X.xty(a, C, pg + HD + D, PGW, BZ, C, D, dW[ENF_W_STEM_W]);               X.colsum(pg + HD + D, PGW, BZ, D, dW[ENF_W_STEM_B]);
  X.colsum(pg + HD + 3 * D, PGW, BZ, D, dW[ENF_W_LNA_G]);                  // sum d an * xn
  X.colsum(pg + HD, PGW, BZ, D, dW[ENF_W_LNA_B]);                          // sum d an
  for (int h = 0; h < H; ++h)                                              // u_h = MU_h k_h: d MU_h[i][dd] = sum d u_h[i] k_h[dd]
    X.xty(dlt + enf_lt_off_u(H, D) + h * D, stride, kv + h * D, 2 * HD, BZ, D, D, dMU + (size_t)h * D * D);
  const float scale = 1.0f / sqrtf((float)m.Dt);                           // ECA:59 (the true width)
  X.colsum(pg + HD + 4 * D, PGW, BZ, HD, dcv, nullptr, 0, scale);          // c_h = cvec_h . k_h; cvec_h = scale (..): dcv = scale d cvec

  // ---- 6. fold backward (enf_pack_weights' folds, enf_layout.h)
  // 6a. MU_h = scale RQ_W2 Q_W[:, h];  cvec_h = scale (RQ_B2 Q_W[:, h] + Q_B[h])
  for (int h = 0; h < H; ++h) {
    const float* qwh = T[ENF_W_Q_W] + h * D;                               // (D x D) slice, leading dimension HD
    X.gemm(dW[ENF_W_RQ_W2], D, dMU + (size_t)h * D * D, D, 0, qwh, HD, 1, D, D, D, scale, h > 0);                 // d W2 += scale dMU_h Qh^T
    X.gemm(dW[ENF_W_Q_W] + h * D, HD, T[ENF_W_RQ_W2], D, 1, dMU + (size_t)h * D * D, D, 0, D, D, D, scale, 0);     // d Qh = scale W2^T dMU_h
    X.scale_outer(dW[ENF_W_Q_W] + h * D, HD, nullptr, 0, nullptr, T[ENF_W_RQ_B2], dcv + h * D, D, D, 1);           //       + b2^T (x) (scale dcvec_h)
    X.gemm(dW[ENF_W_RQ_B2], D, dcv + h * D, D, 0, qwh, HD, 1, 1, D, D, 1.f, h > 0);                               // d b2 += (scale dcvec_h) Qh^T
  }
  X.axpy(dW[ENF_W_Q_B], dcv, HD);
  // 6b. AF = RV_W2 F1_W0;  bf = RV_B2 F1_W0 + F1_B0
  const float* dAF = dpair[ENF_P_AF]; const float* dBF = dpair[ENF_P_BF];
  X.gemm(dW[ENF_W_RV_W2], D, dAF, D, 0, T[ENF_W_F1_W0], D, 1, D, D, D);
  X.gemm(dW[ENF_W_F1_W0], D, T[ENF_W_RV_W2], D, 1, dAF, D, 0, D, D, D);
  X.scale_outer(dW[ENF_W_F1_W0], D, nullptr, 0, nullptr, T[ENF_W_RV_B2], dBF, D, D, 1);
  X.gemm(dW[ENF_W_RV_B2], D, dBF, D, 0, T[ENF_W_F1_W0], D, 1, 1, D, D);
  X.axpy(dW[ENF_W_F1_B0], dBF, D);
  // 6c. AGB = diag(F1_G) F1_W1;  bgb = F1_BE F1_W1 + F1_B1       (D x 2HD)
  const float* dAGB = dpair[ENF_P_AGB]; const float* dBGB = dpair[ENF_P_BGB];
  X.rowdot(dW[ENF_W_F1_G], dAGB, 2 * HD, T[ENF_W_F1_W1], 2 * HD, D, 2 * HD);
  X.scale_outer(dW[ENF_W_F1_W1], 2 * HD, dAGB, 2 * HD, T[ENF_W_F1_G], T[ENF_W_F1_BE], dBGB, D, 2 * HD);
  X.gemm(dW[ENF_W_F1_BE], D, dBGB, 2 * HD, 0, T[ENF_W_F1_W1], 2 * HD, 1, 1, D, 2 * HD);
  X.axpy(dW[ENF_W_F1_B1], dBGB, 2 * HD);
  // 6d. direct per-pair tensors
  X.axpy(dW[ENF_W_RQ_W1], dpair[ENF_P_AQ1], D * D); X.axpy(dW[ENF_W_RQ_B1], dpair[ENF_P_BQ1], D);
  X.axpy(dW[ENF_W_RV_W1], dpair[ENF_P_AV1], D * D); X.axpy(dW[ENF_W_RV_B1], dpair[ENF_P_BV1], D);
  X.axpy(dW[ENF_W_MX_W0], dpair[ENF_P_AM], D * D);  X.axpy(dW[ENF_W_MX_B0], dpair[ENF_P_BM], D);
  // 6e. WB_h = mxw T_h,  T = AO_W FF_W0,  bB = AO_B FF_W0 + FF_B0 + sum_h mxb T_h;   mxw = diag(MX_G) MX_W1, mxb = MX_BE MX_W1 + MX_B1
  X.gemm(Tm, HD, T[ENF_W_AO_W], HD, 0, T[ENF_W_FF_W0], HD, 0, HD, HD, HD);
  for (int h = 0; h < H; ++h) {
    const float* Th = Tm + (size_t)h * D * HD;                             // rows hD .. (h+1)D of T: D x HD
    const float* dWBh = dWB + (size_t)h * D * HD;
    X.gemm(dmxw, D, dWBh, HD, 0, Th, HD, 1, D, D, HD, 1.f, h > 0);                                 // d mxw += dWB_h T_h^T
    X.gemm(dmxb, D, dbB, HD, 0, Th, HD, 1, 1, D, HD, 1.f, h > 0);                                  // d mxb += dbB T_h^T
    X.gemm(dT + (size_t)h * D * HD, HD, Bf(L.p_mxw), D, 1, dWBh, HD, 0, D, HD, D);                 // d T_h = mxw^T dWB_h
    X.scale_outer(dT + (size_t)h * D * HD, HD, nullptr, 0, nullptr, Bf(L.p_mxb), dbB, D, HD, 1);   //        + mxb^T (x) dbB
  }
  X.gemm(dW[ENF_W_AO_W], HD, dT, HD, 0, T[ENF_W_FF_W0], HD, 1, HD, HD, HD);
  X.gemm(dW[ENF_W_FF_W0], HD, T[ENF_W_AO_W], HD, 1, dT, HD, 0, HD, HD, HD);
  X.scale_outer(dW[ENF_W_FF_W0], HD, nullptr, 0, nullptr, T[ENF_W_AO_B], dbB, HD, HD, 1);
  X.gemm(dW[ENF_W_AO_B], HD, dbB, HD, 0, T[ENF_W_FF_W0], HD, 1, 1, HD, HD);
  X.axpy(dW[ENF_W_FF_B0], dbB, HD);
  X.rowdot(dW[ENF_W_MX_G], dmxw, D, T[ENF_W_MX_W1], D, D, D);
  X.scale_outer(dW[ENF_W_MX_W1], D, dmxw, D, T[ENF_W_MX_G], T[ENF_W_MX_BE], dmxb, D, D);
  X.gemm(dW[ENF_W_MX_BE], D, dmxb, D, 0, T[ENF_W_MX_W1], D, 1, 1, D, D);
  X.axpy(dW[ENF_W_MX_B1], dmxb, D);
  // 6f. wf1 = diag(FF_G) FF_W1;  bF1 = FF_BE FF_W1 + FF_B1
  X.rowdot(dW[ENF_W_FF_G], dwf1, HD, T[ENF_W_FF_W1], HD, HD, HD);
  X.scale_outer(dW[ENF_W_FF_W1], HD, dwf1, HD, T[ENF_W_FF_G], T[ENF_W_FF_BE], dbF1, HD, HD);
  X.gemm(dW[ENF_W_FF_BE], HD, dbF1, HD, 0, T[ENF_W_FF_W1], HD, 1, 1, HD, HD);
  X.axpy(dW[ENF_W_FF_B1], dbF1, HD);
  // frozen RFF coefficients (rff.py:87-90): zero gradient where the caller asked for one
  if (dW[ENF_W_RQ_COEF] && hipMemsetAsync(dW[ENF_W_RQ_COEF], 0, sizeof(float) * (size_t)m.I * (D / 2), st) != hipSuccess) return ENF_ELAUNCH;
  if (dW[ENF_W_RV_COEF] && hipMemsetAsync(dW[ENF_W_RV_COEF], 0, sizeof(float) * (size_t)m.I * (D / 2), st) != hipSuccess) return ENF_ELAUNCH;
  return X.rc;
}
