// enf_pack.hip -- K0 (weight packing + exact folds) and K1 (latent prologue, forward + backward).
//
// K0 turns the Flax weight tree of EquivariantCrossAttentionNeF into the packed
// blob described in enf_layout.h.  K1 is the per-latent part of the decoder:
//   latent_stem (NEF:220) -> layer_norm_attn (NEF:56) -> a_to_k / a_to_v (ECA:93-94)
// followed by the fold of inv_emb_to_q into the keys (ECA:92 + ECA:134):
//   att[n,z,h] = scale * q[n,z,h,:] . k[z,h,:]
//              = h1[n,z,:] . u[z,h,:] + c[z,h],   u = scale * W2 Wq_h k_h,  c = scale * (b2 Wq_h + bq_h) . k_h
// where h1 is the relu layer of the query RFFNet and (W2, b2) its linear_final (RFF:46).
#include <hip/hip_runtime.h>
#include <math.h>
#include "enf_layout.h"

#define CK(x) do { if ((x) != hipSuccess) return ENF_ELAUNCH; } while (0)

// ---------------------------------------------------------------- small dense helpers (fp32)
// C[i][j] = (acc ? C[i][j] : 0) + alpha * (sum_k A[i*lda+k] * B[k*ldb+j] + (addv ? addv[j] : 0))
__global__ __launch_bounds__(256) void mm_kernel(float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K,
                                                 float alpha, const float* addv, int acc) {
  // 16 x 16 output tile per workgroup, operands through LDS (the one-thread-per-output form read A uncoalesced: 32 us per call)
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
  float s = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    sa[ty][tx] = (i < M && k0 + tx < K) ? A[(size_t)i * lda + k0 + tx] : 0.f;
    sb[ty][tx] = (k0 + ty < K && j < N) ? B[(size_t)(k0 + ty) * ldb + j] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) s = fmaf(sa[ty][k], sb[k][tx], s);
    __syncthreads();
  }
  if (i >= M || j >= N) return;
  if (addv) s += addv[j];
  s *= alpha;
  C[(size_t)i * ldc + j] = acc ? C[(size_t)i * ldc + j] + s : s;
}
// dst[i][j] = g[i] * src[i][j]
__global__ void rowscale_kernel(float* dst, const float* src, const float* g, int rows, int cols) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j < cols && i < rows) dst[(size_t)i * cols + j] = g[i] * src[(size_t)i * cols + j];
}
// gamma/beta column reorder (ECA:115 split: gamma = first HD columns, beta = last HD): per head h the
// 32-wide blocks alternate [g_h blk0 | b_h blk0 | g_h blk1 | b_h blk1 ..] so that one staged slice of the
// panel holds gamma AND beta of the same features:
//   dst col h*2D + (2m+t)*32 + i  <-  src col t*HD + h*D + 32m + i      (t: 0 gamma, 1 beta); rows scaled by g
__global__ void reorder_gb_kernel(float* dst, const float* src, const float* g, int rows, int H, int D) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  const int HD = H * D;
  if (jj >= 2 * HD || i >= rows) return;
  const int h = jj / (2 * D), w = jj % (2 * D), blk = w / 32, ii = w % 32, m = blk >> 1, t = blk & 1;
  dst[(size_t)i * 2 * HD + jj] = (g ? g[i] : 1.f) * src[(size_t)i * 2 * HD + t * HD + h * D + 32 * m + ii];
}
// dst (cols x rows) = src (rows x cols)^T
__global__ void transpose_kernel(float* dst, const float* src, int rows, int cols) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j < cols && i < rows) dst[(size_t)j * rows + i] = src[(size_t)i * cols + j];
}
// zero-padded copy of a (rows x cols) matrix into (rows x cols_pad)
__global__ void padcopy_kernel(float* dst, const float* src, int rows, int cols, int cols_pad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j < cols_pad && i < rows) dst[(size_t)i * cols_pad + j] = j < cols ? src[(size_t)i * cols + j] : 0.f;
}
// RFF coefficient A-operand of t = coeff^T inv (v_mfma_f32_16x16x4_f32, K = 4 = the I <= 4 invariant
// components): [t-tile][lane], lane (i = lane&15, q = lane>>4) holds coeff[q][16 tt + i]
__global__ void coef_frag_kernel(float* dst, const float* coeff, int I, int Dh /*D/2*/) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (Dh / 16) * 64;
  if (idx >= total) return;
  const int lane = idx & 63, tt = idx >> 6;
  const int c = lane >> 4, t = 16 * tt + (lane & 15);
  dst[idx] = c < I ? coeff[(size_t)c * Dh + t] : 0.f;
}

// coefficient rows in kernel order: dst (8 x Dh) = [the <= 4 per-pair rows | the <= 2 latent-only rows | 0] (enf_inv_rows)
__global__ void coef_perm_kernel(float* dst, const float* coeff, int inv, int I, int Dh) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (t >= Dh) return;
  int pair[4], lat[2], np, nl;
  enf_inv_rows(inv, I, pair, np, lat, nl);
  const int src = r < 4 ? (r < np ? pair[r] : -1) : (r - 4 < nl ? lat[r - 4] : -1);
  dst[(size_t)r * Dh + t] = src >= 0 ? coeff[(size_t)src * Dh + t] : 0.f;
}

// ---------------------------------------------------------------- MFMA A-operand panel packer
// A[r][k] (R x K; R multiple of 16, K multiple of 32) = scale * (trans ? W[r*ldw + k] : W[k*ldw + r]).
// bf16 (v_mfma_f32_16x16x32_bf16): byte ((mt*KB+blk)*64+lane)*16 + 2j
//        <- A[16mt + (lane&15)][32blk + 16(j>>2) + 4(lane>>4) + (j&3)]
// fp32 (v_mfma_f32_16x16x4_f32):   byte ((mt*2KB+tin)*64+lane)*16 + 4i
//        <- A[16mt + (lane&15)][16tin + 4(lane>>4) + i]
// (the k order is the one in which 16x16 accumulator tiles present their rows as the next B operand)
__global__ void pack_panel_kernel(void* dst, const float* W, int ldw, int R, int K, int trans, int bf16,
                                  int Rvalid, int Kvalid, float scale) {
  const size_t total = (size_t)R * K;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int r, k;
  if (bf16) {
    const int KB = K / 32;
    const int j = e & 7, lane = (e >> 3) & 63;
    const size_t g = e >> 9;  // mt*KB + blk
    const int blk = g % KB, mt = g / KB;
    r = 16 * mt + (lane & 15);
    k = 32 * blk + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
  } else {
    const int KT = K / 16;
    const int i = e & 3, lane = (e >> 2) & 63;
    const size_t g = e >> 8;  // mt*KT + tin
    const int tin = g % KT, mt = g / KT;
    r = 16 * mt + (lane & 15);
    k = 16 * tin + 4 * (lane >> 4) + i;
  }
  float v = 0.f;
  if (r < Rvalid && k < Kvalid) v = scale * (trans ? W[(size_t)r * ldw + k] : W[(size_t)k * ldw + r]);
  if (bf16) reinterpret_cast<__bf16*>(dst)[e] = (__bf16)v;
  else reinterpret_cast<float*>(dst)[e] = v;
}

// gamma half of one head out of the [g g b b]-interleaved D x 2HD matrix: dst[i][j] = Wgamma_h[i][j]
__global__ void gamma_extract_kernel(float* dst, const float* agb, int h, int H, int D) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j < D) dst[(size_t)i * D + j] = agb[(size_t)i * 2 * H * D + h * 2 * D + 64 * (j >> 5) + (j & 31)];
}

// latent-independent parts of the per-latent fold W_zh = (Wgamma_h diag(v0) + Wbeta_h) AM (enf_wz.hip):
//   wbmt[h][k][i] = sum_j Wbeta_h[i][j] AM[j][k];  cb[h][k] = sum_j bbeta_h[j] AM[j][k] + bm[k];  opbg[h][j] = 1 + bgamma_h[j]
// agb / bgb are in the [g g b b] interleaved column order of reorder_gb_kernel.
__global__ void wz_const_kernel(float* wbmt, float* wbm, float* cb, float* opbg, const float* agb, const float* bgb, const float* am,
                                const float* bm, int H, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y, h = blockIdx.z;
  if (i >= D) return;
  const int HD2 = 2 * H * D;
  auto bcol = [&](int j) { return h * 2 * D + 64 * (j >> 5) + 32 + (j & 31); };   // beta column of feature j
  float s = 0.f;
  for (int j = 0; j < D; ++j) s = fmaf(agb[(size_t)i * HD2 + bcol(j)], am[(size_t)j * D + k], s);
  wbmt[((size_t)h * D + k) * D + i] = s;
  wbm[((size_t)h * D + i) * D + k] = s;
  if (i == 0) {
    float c = bm[k];
    for (int j = 0; j < D; ++j) c = fmaf(bgb[bcol(j)], am[(size_t)j * D + k], c);
    cb[h * D + k] = c;
  }
  if (k == 0) opbg[h * D + i] = 1.0f + bgb[h * 2 * D + 64 * (i >> 5) + (i & 31)];
}

static inline int mm(hipStream_t st, float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N,
                     int K, float alpha, const float* addv, int acc) {
  dim3 g((N + 15) / 16, (M + 15) / 16);
  hipLaunchKernelGGL(mm_kernel, g, dim3(256), 0, st, C, ldc, A, lda, B, ldb, M, N, K, alpha, addv, acc);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}
static inline int pack_panel(hipStream_t st, char* blob, size_t off, const float* W, int ldw, int R, int K, int trans,
                             int bf16, int Rvalid = -1, int Kvalid = -1, float scale = 1.0f) {
  const size_t total = (size_t)R * K;
  hipLaunchKernelGGL(pack_panel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (void*)(blob + off), W,
                     ldw, R, K, trans, bf16, Rvalid < 0 ? R : Rvalid, Kvalid < 0 ? K : Kvalid, scale);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

// Panels of the per-pair chain (forward: A[out][in] = W[in][out]; backward: A[in][out] = W[in][out], dX = W dY)
// from plain (in, out) matrices; `agb` is D x 2HD in the [g g b b] interleaved column order.
static int pack_pair_panels(hipStream_t st, char* blob, const EnfLayout& L, const EnfDims& m, const float* aq1,
                            const float* av1, const float* af, const float* agb, const float* am, const float* coefq,
                            const float* coefv) {
  const int D = m.D, H = m.H, HD = m.HD, I = m.I, bf = m.bf16;
  int rc;
  {
    // coefficient rows permuted to [per-pair rows | latent-only rows] (identity except for ball / ball_lat)
    float* cq = reinterpret_cast<float*>(blob + L.p_tmp);
    float* cv = cq + 8 * (D / 2);
    hipLaunchKernelGGL(coef_perm_kernel, dim3((D / 2 + 63) / 64, 8), dim3(64), 0, st, cq, coefq, m.inv, I, D / 2);
    hipLaunchKernelGGL(coef_perm_kernel, dim3((D / 2 + 63) / 64, 8), dim3(64), 0, st, cv, coefv, m.inv, I, D / 2);
    const int tot = (D / 32) * 64;
    hipLaunchKernelGGL(coef_frag_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, reinterpret_cast<float*>(blob + L.acq), cq, 4, D / 2);
    hipLaunchKernelGGL(coef_frag_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, reinterpret_cast<float*>(blob + L.acv), cv, 4, D / 2);
    if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
    if (hipMemcpyAsync(blob + L.cphq, cq + 4 * (D / 2), sizeof(float) * D, hipMemcpyDeviceToDevice, st) != hipSuccess) return ENF_ELAUNCH;
    if (hipMemcpyAsync(blob + L.cphv, cv + 4 * (D / 2), sizeof(float) * D, hipMemcpyDeviceToDevice, st) != hipSuccess) return ENF_ELAUNCH;
    // d inv[c] = sum_t 2 pi coeff[c][t] d t[t]  (RFF:92), rows in the same order (latent-only rows 4, 5)
    if ((rc = pack_panel(st, blob, L.gcq, cq, D / 2, 16, D / 2, 1, bf, 8, D / 2, 6.283185307179586f))) return rc;
    if ((rc = pack_panel(st, blob, L.gcv, cv, D / 2, 16, D / 2, 1, bf, 8, D / 2, 6.283185307179586f))) return rc;
  }
  if ((rc = pack_panel(st, blob, L.aq1, aq1, D, D, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.av1, av1, D, D, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.af, af, D, D, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.agb, agb, 2 * HD, 2 * HD, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.am, am, D, D, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gq1, aq1, D, D, D, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gv1, av1, D, D, D, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gf, af, D, D, D, 1, bf))) return rc;
  for (int h = 0; h < H; ++h)   // one K-slice (that head's [g g b b ..] 2D columns) per head
    if ((rc = pack_panel(st, blob, L.ggb + (size_t)h * enf_panel_bytes(D, 2 * D, bf), agb + h * 2 * D, 2 * HD, D, 2 * D, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gm, am, D, D, D, 1, bf))) return rc;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(blob + off); };
  hipLaunchKernelGGL(wz_const_kernel, dim3((D + 63) / 64, D, H), dim3(64), 0, st, F(L.p_wbmt), F(L.p_wbm), F(L.p_cb), F(L.p_opbg), agb,
                     F(L.bgb), am, F(L.bm), H, D);
  if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
  for (int h = 0; h < H; ++h) {      // gamma-only forward panels (z-fold backward: flipped 1 + gamma for d v0)
    hipLaunchKernelGGL(gamma_extract_kernel, dim3((D + 63) / 64, D), dim3(64), 0, st, F(L.p_tmp), agb, h, H, D);
    if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
    if ((rc = pack_panel(st, blob, L.awg + (size_t)h * enf_panel_bytes(D, D, bf), F(L.p_tmp), D, D, D, 0, bf))) return rc;
  }
  return ENF_OK;
}

// Pack ONLY what the pair kernels (enf_pair_forward / enf_pair_backward) read, from the "effective"
// per-pair parameters (ENF_P_* order, include/enf_hip.h): used by the training path, where folds,
// latent prologue and tail run as differentiable host-framework ops around the HIP pair kernels.
extern "C" int enf_pack_pair(const EnfDesc* d, const float* const* T, void* packed, void* stream) {
  if (!d || !T || !packed) return ENF_EINVAL;
  int rc = enf_check_desc(d);
  if (rc) return rc;
  for (int i = 0; i < ENF_NUM_PAIR_TENSORS; ++i)
    if (!T[i]) return ENF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const EnfDims m = enf_dims(d);
  const EnfLayout L = enf_layout(m);
  char* blob = (char*)packed;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(blob + off); };
  const int D = m.D, H = m.H, HD = m.HD;
  const size_t f = sizeof(float);
  auto cp = [&](size_t off, const float* src, size_t n) {
    return hipMemcpyAsync(blob + off, src, n * f, hipMemcpyDeviceToDevice, st);
  };
  CK(cp(L.bq1, T[ENF_P_BQ1], D)); CK(cp(L.bv1, T[ENF_P_BV1], D)); CK(cp(L.bf, T[ENF_P_BF], D)); CK(cp(L.bm, T[ENF_P_BM], D));
  hipLaunchKernelGGL(reorder_gb_kernel, dim3((2 * HD + 127) / 128, 1), dim3(128), 0, st, F(L.bgb), T[ENF_P_BGB], (const float*)nullptr, 1, H, D);
  hipLaunchKernelGGL(reorder_gb_kernel, dim3((2 * HD + 127) / 128, D), dim3(128), 0, st, F(L.p_agb), T[ENF_P_AGB], (const float*)nullptr, D, H, D);
  if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
  return pack_pair_panels(st, blob, L, m, T[ENF_P_AQ1], T[ENF_P_AV1], T[ENF_P_AF], F(L.p_agb), T[ENF_P_AM], T[ENF_P_COEFQ],
                          T[ENF_P_COEFV]);
}

#ifdef ENF_TEST_HOOKS
// test-only build (libenf_hip_test.so): pack a plain fp32 (K x M row-major, W[k][m]) matrix as the A operand A[m][k] = W[k][m]
extern "C" int enf_debug_pack(void* dst, const float* W, int M, int K, int bf16, void* stream) {
  const size_t total = (size_t)M * K;
  hipLaunchKernelGGL(pack_panel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, W, M,
                     M, K, 0, bf16, M, K, 1.0f);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}
#endif

extern "C" int enf_pack_weights(const EnfDesc* d, const float* const* T, void* packed, void* stream) {
  if (!d || !T || !packed) return ENF_EINVAL;
  int rc = enf_check_desc(d);
  if (rc) return rc;
  for (int i = 0; i < ENF_NUM_TENSORS; ++i)
    if (!T[i]) return ENF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const EnfDims m = enf_dims(d);
  const EnfLayout L = enf_layout(m);
  char* blob = (char*)packed;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(blob + off); };
  const int D = m.D, H = m.H, HD = m.HD, C = m.C, O = m.O, OP = 32 * m.OB;
  const size_t f = sizeof(float);
  const float scale = 1.0f / sqrtf((float)m.Dt);  // ECA:59 (num_hidden, not a padded width)
  auto cp = [&](size_t off, const float* src, size_t n) {
    return hipMemcpyAsync(blob + off, src, n * f, hipMemcpyDeviceToDevice, st);
  };
  // ---- prologue tensors
  CK(cp(L.stem_w, T[ENF_W_STEM_W], (size_t)C * D)); CK(cp(L.stem_b, T[ENF_W_STEM_B], D));
  CK(cp(L.lna_g, T[ENF_W_LNA_G], D)); CK(cp(L.lna_b, T[ENF_W_LNA_B], D));
  CK(cp(L.wk, T[ENF_W_K_W], (size_t)D * HD)); CK(cp(L.bk, T[ENF_W_K_B], HD));
  CK(cp(L.wv, T[ENF_W_V_W], (size_t)D * HD)); CK(cp(L.bv, T[ENF_W_V_B], HD));
  for (int h = 0; h < H; ++h) {
    if ((rc = mm(st, F(L.mu) + (size_t)h * D * D, D, T[ENF_W_RQ_W2], D, T[ENF_W_Q_W] + h * D, HD, D, D, D, scale, nullptr, 0))) return rc;
    if ((rc = mm(st, F(L.cvec) + (size_t)h * D, D, T[ENF_W_RQ_B2], D, T[ENF_W_Q_W] + h * D, HD, 1, D, D, scale, T[ENF_W_Q_B] + h * D, 0))) return rc;
  }
  for (int h = 0; h < H; ++h)
    hipLaunchKernelGGL(transpose_kernel, dim3((D + 127) / 128, D), dim3(128), 0, st, F(L.mut) + (size_t)h * D * D,
                       F(L.mu) + (size_t)h * D * D, D, D);
  hipLaunchKernelGGL(transpose_kernel, dim3((HD + 127) / 128, D), dim3(128), 0, st, F(L.wkt), T[ENF_W_K_W], D, HD);
  hipLaunchKernelGGL(transpose_kernel, dim3((HD + 127) / 128, D), dim3(128), 0, st, F(L.wvt), T[ENF_W_V_W], D, HD);
  if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
  // ---- per-pair biases
  CK(cp(L.bq1, T[ENF_W_RQ_B1], D)); CK(cp(L.bv1, T[ENF_W_RV_B1], D)); CK(cp(L.bm, T[ENF_W_MX_B0], D));
  if ((rc = mm(st, F(L.bf), D, T[ENF_W_RV_B2], D, T[ENF_W_F1_W0], D, 1, D, D, 1.f, T[ENF_W_F1_B0], 0))) return rc;
  // bgb = LN.bias @ Dense_1 + Dense_1.bias, reordered like the panel (see reorder_gb_kernel)
  if ((rc = mm(st, F(L.p_tmp), 2 * HD, T[ENF_W_F1_BE], D, T[ENF_W_F1_W1], 2 * HD, 1, 2 * HD, D, 1.f, T[ENF_W_F1_B1], 0))) return rc;
  hipLaunchKernelGGL(reorder_gb_kernel, dim3((2 * HD + 127) / 128, 1), dim3(128), 0, st, F(L.bgb), F(L.p_tmp), (const float*)nullptr, 1, H, D);
  // ---- folded plain matrices
  if ((rc = mm(st, F(L.p_af), D, T[ENF_W_RV_W2], D, T[ENF_W_F1_W0], D, D, D, D, 1.f, nullptr, 0))) return rc;
  hipLaunchKernelGGL(reorder_gb_kernel, dim3((2 * HD + 127) / 128, D), dim3(128), 0, st, F(L.p_agb), T[ENF_W_F1_W1], T[ENF_W_F1_G], D, H, D);
  hipLaunchKernelGGL(rowscale_kernel, dim3((D + 127) / 128, D), dim3(128), 0, st, F(L.p_mxw), T[ENF_W_MX_W1], T[ENF_W_MX_G], D, D);
  if ((rc = mm(st, F(L.p_mxb), D, T[ENF_W_MX_BE], D, T[ENF_W_MX_W1], D, 1, D, D, 1.f, T[ENF_W_MX_B1], 0))) return rc;
  if ((rc = mm(st, F(L.p_tmp), HD, T[ENF_W_AO_W], HD, T[ENF_W_FF_W0], HD, HD, HD, HD, 1.f, nullptr, 0))) return rc;
  if ((rc = mm(st, F(L.bB), HD, T[ENF_W_AO_B], HD, T[ENF_W_FF_W0], HD, 1, HD, HD, 1.f, T[ENF_W_FF_B0], 0))) return rc;
  for (int h = 0; h < H; ++h) {
    if ((rc = mm(st, F(L.p_wb) + (size_t)h * D * HD, HD, F(L.p_mxw), D, F(L.p_tmp) + (size_t)h * D * HD, HD, D, HD, D, 1.f, nullptr, 0))) return rc;
    if ((rc = mm(st, F(L.bB), HD, F(L.p_mxb), D, F(L.p_tmp) + (size_t)h * D * HD, HD, 1, HD, D, 1.f, nullptr, 1))) return rc;
  }
  hipLaunchKernelGGL(rowscale_kernel, dim3((HD + 127) / 128, HD), dim3(128), 0, st, F(L.p_wf1), T[ENF_W_FF_W1], T[ENF_W_FF_G], HD, HD);
  if ((rc = mm(st, F(L.bF1), HD, T[ENF_W_FF_BE], HD, T[ENF_W_FF_W1], HD, 1, HD, HD, 1.f, T[ENF_W_FF_B1], 0))) return rc;
  CK(cp(L.bO0, T[ENF_W_O0_B], D)); CK(cp(L.bO2, T[ENF_W_O2_B], D));
  hipLaunchKernelGGL(padcopy_kernel, dim3((OP + 127) / 128, 1), dim3(128), 0, st, F(L.bO4), T[ENF_W_O4_B], 1, O, OP);
  hipLaunchKernelGGL(padcopy_kernel, dim3((OP + 127) / 128, D), dim3(128), 0, st, F(L.p_o4), T[ENF_W_O4_W], D, O, OP);
  if (hipGetLastError() != hipSuccess) return ENF_ELAUNCH;
  // ---- panels
  const int bf = m.bf16;
  if ((rc = pack_pair_panels(st, blob, L, m, T[ENF_W_RQ_W1], T[ENF_W_RV_W1], F(L.p_af), F(L.p_agb), T[ENF_W_MX_W0],
                             T[ENF_W_RQ_COEF], T[ENF_W_RV_COEF]))) return rc;
  // tail, forward: A[out][in] = W[in][out]; backward: A[in][out] = W[in][out]  (dX = W dY)
  if ((rc = pack_panel(st, blob, L.atb, F(L.p_wb), HD, HD, HD, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.atf1, F(L.p_wf1), HD, HD, HD, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.ato0, T[ENF_W_O0_W], D, D, HD, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.ato2, T[ENF_W_O2_W], D, D, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.ato4, F(L.p_o4), OP, OP, D, 0, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gtb, F(L.p_wb), HD, HD, HD, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gtf1, F(L.p_wf1), HD, HD, HD, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gto0, T[ENF_W_O0_W], D, HD, D, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gto2, T[ENF_W_O2_W], D, D, D, 1, bf))) return rc;
  if ((rc = pack_panel(st, blob, L.gto4, F(L.p_o4), OP, D, OP, 1, bf))) return rc;
  return ENF_OK;
}

// ---------------------------------------------------------------- K1 latent prologue
// One block = ZT consecutive (b,z) rows, 256 threads.  Every weight matrix is read with the OUTPUT index on the
// lanes (coalesced: stem_w, wk, wv are (in,out) row-major; the logit fold uses the transposed copy muT) and every
// weight element is read once per ZT latents.  Activations sit in LDS as [feature][latent] so that one
// ds_read_b128 pair broadcasts the ZT values a weight element multiplies.  LayerNorm: one wave per row.
#ifndef ENF_PROLOGUE_ZT
#define ENF_PROLOGUE_ZT 4
#endif
constexpr int ZT = ENF_PROLOGUE_ZT;   // latents per block (multiple of 4)
#ifndef ENF_PROLOGUE_UNROLL
#define ENF_PROLOGUE_UNROLL 16         // independent weight loads in flight per thread: these kernels are latency chains
#endif
#define PRO_UNROLL ENF_PROLOGUE_UNROLL
typedef float pf4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct PrologueArgs {
  const float* p; const float* a; const float* sigma;
  const char* blob; EnfLayout L;
  float* lt; float* an; float* kv;
  int BZ, H, D, C, dp, inv;
  int Dt;                // true num_hidden (LayerNorm statistics); D may be a zero-padded width
};

__global__ __launch_bounds__(256) void enf_prologue_kernel(PrologueArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = A.D, H = A.H, HD = H * D, C = A.C;
  float* s_a = sm;                 // [C][ZT]
  float* s_an = s_a + ZT * C;      // [D][ZT]   stem output, then the normalised + affine row
  float* s_k = s_an + ZT * D;      // [HD][ZT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * ZT;
  auto W = [&](size_t off) { return reinterpret_cast<const float*>(A.blob + off); };
  const int stride = enf_lt_stride(H, D);
  for (int i = tid; i < ZT * C; i += 256) {
    const int zz = i / C, c = i % C, r = row0 + zz;
    s_a[c * ZT + zz] = r < A.BZ ? A.a[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
  // stem: s = a @ Ws + bs  (NEF:220)
  for (int d = tid; d < D; d += 256) {
    float acc[ZT];
    const float bs = W(A.L.stem_b)[d];
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) acc[zz] = bs;
#pragma unroll PRO_UNROLL
    for (int c = 0; c < C; ++c) {
      const float w = W(A.L.stem_w)[(size_t)c * D + d];
#pragma unroll
      for (int q = 0; q < ZT / 4; ++q) {
        const pf4 x = *reinterpret_cast<const pf4*>(s_a + c * ZT + 4 * q);
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[4 * q + zz] = fmaf(x[zz], w, acc[4 * q + zz]);
      }
    }
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) s_an[d * ZT + zz] = acc[zz];
  }
  __syncthreads();
  // LayerNorm (NEF:56): biased variance, eps 1e-6; one wave per row
  for (int zz = wave; zz < ZT; zz += 4) {
    const int r = row0 + zz;
    float v = 0.f;
    for (int d = lane; d < A.Dt; d += 64) v += s_an[d * ZT + zz];          // statistics over the real features only
    const float mu = wave_sum(v) / A.Dt;
    float q = 0.f;
    for (int d = lane; d < A.Dt; d += 64) { const float t = s_an[d * ZT + zz] - mu; q += t * t; }
    const float rstd = rsqrtf(wave_sum(q) / A.Dt + 1e-6f);
    for (int d = lane; d < D; d += 64) {
      const float sv = s_an[d * ZT + zz];
      const float xn = (sv - mu) * rstd;
      if (r < A.BZ) {
        A.an[(size_t)r * (2 * D + 2) + d] = sv;          // stem output (pre-LN)
        A.an[(size_t)r * (2 * D + 2) + D + d] = xn;      // normalised, before scale/bias
      }
      s_an[d * ZT + zz] = xn * W(A.L.lna_g)[d] + W(A.L.lna_b)[d];
    }
    if (lane == 0 && r < A.BZ) { A.an[(size_t)r * (2 * D + 2) + 2 * D] = mu; A.an[(size_t)r * (2 * D + 2) + 2 * D + 1] = rstd; }
  }
  __syncthreads();
  // k = an @ Wk + bk, v0 = an @ Wv + bv   (ECA:93-94)
  for (int j = tid; j < 2 * HD; j += 256) {
    const bool isv = j >= HD;
    const int jj = isv ? j - HD : j;
    const float* Wm = W(isv ? A.L.wv : A.L.wk);
    float acc[ZT];
    const float bb = W(isv ? A.L.bv : A.L.bk)[jj];
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) acc[zz] = bb;
#pragma unroll PRO_UNROLL
    for (int d = 0; d < D; ++d) {
      const float w = Wm[(size_t)d * HD + jj];
#pragma unroll
      for (int q = 0; q < ZT / 4; ++q) {
        const pf4 x = *reinterpret_cast<const pf4*>(s_an + d * ZT + 4 * q);
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[4 * q + zz] = fmaf(x[zz], w, acc[4 * q + zz]);
      }
    }
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) {
      const int r = row0 + zz;
      if (!isv) s_k[jj * ZT + zz] = acc[zz];
      if (r < A.BZ) {
        A.kv[(size_t)r * 2 * HD + j] = acc[zz];
        if (isv) A.lt[(size_t)r * stride + enf_lt_off_v0(H, D) + jj] = acc[zz];
      }
    }
  }
  __syncthreads();
  // u_h = MU_h k_h  (muT[h][d][i]: the output index i on the lanes)
  for (int j = tid; j < HD; j += 256) {
    const int h = j / D, i = j % D;
    const float* mut = W(A.L.mut) + (size_t)h * D * D + i;
    float acc[ZT];
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) acc[zz] = 0.f;
#pragma unroll PRO_UNROLL
    for (int dd = 0; dd < D; ++dd) {
      const float w = mut[(size_t)dd * D];
      const float* kp = s_k + (h * D + dd) * ZT;
#pragma unroll
      for (int q = 0; q < ZT / 4; ++q) {
        const pf4 x = *reinterpret_cast<const pf4*>(kp + 4 * q);
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[4 * q + zz] = fmaf(x[zz], w, acc[4 * q + zz]);
      }
    }
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) {
      const int r = row0 + zz;
      if (r < A.BZ) A.lt[(size_t)r * stride + enf_lt_off_u(H, D) + j] = acc[zz];
    }
  }
  // c_h = cvec_h . k_h : one wave per (row, head)
  for (int t = wave; t < ZT * H; t += 4) {
    const int zz = t / H, h = t % H, r = row0 + zz;
    float sacc = 0.f;
    for (int dd = lane; dd < D; dd += 64) sacc = fmaf(W(A.L.cvec)[h * D + dd], s_k[(h * D + dd) * ZT + zz], sacc);
    sacc = wave_sum(sacc);
    if (lane == 0 && r < A.BZ) A.lt[(size_t)r * stride + enf_lt_off_c(H, D) + h] = sacc;
  }
  // pose embed (NEF:214-217) + window coefficient
  if (tid < ZT) {
    const int r = row0 + tid;
    if (r < A.BZ) {
      const float* pp = A.p + (size_t)r * A.dp;
      float q[4] = {0.f, 0.f, 0.f, 0.f};
      float wc;
      const float sg = A.sigma ? A.sigma[r] : 1.f;
      const bool sphere = A.inv == ENF_INV_LATITUDE_PERIODIC || A.inv == ENF_INV_POLAR_PERIODIC || enf_inv_has_phase(A.inv);
      if (A.inv == ENF_INV_PONITA || A.inv == ENF_INV_PONITA_FULL) { q[0] = pp[0]; q[1] = pp[1]; q[2] = cosf(pp[2]); q[3] = sinf(pp[2]); }
      else if (sphere) { q[0] = pp[0]; q[1] = pp[1]; q[2] = sinf(pp[1]); q[3] = cosf(pp[1]); }    // ball: (alpha, beta) play (phi, theta) in the window
      else { for (int i = 0; i < A.dp && i < 3; ++i) q[i] = pp[i]; }
      wc = sphere ? 1.f / (2.f * sg * sg) : 1.f / (sg * sg);
      float* o = A.lt + (size_t)r * stride;
      for (int i = 0; i < 4; ++i) o[enf_lt_off_pose(H, D) + i] = q[i];
      o[enf_lt_off_wcoef(H, D)] = wc;
      if (enf_inv_has_phase(A.inv)) {
        float lat[2] = {0.f, 0.f};
        if (A.inv == ENF_INV_BALL) {            // R(alpha, beta, gamma), ball.py:76-84; latent-only invariant r_p
          const float ca = cosf(pp[0]), sa = sinf(pp[0]), cb = cosf(pp[1]), sb = sinf(pp[1]), cg = cosf(pp[2]), sg2 = sinf(pp[2]);
          float* R = o + enf_lt_off_ext(H, D);
          R[0] = ca * cb; R[1] = ca * sb * sg2 - sa * cg; R[2] = ca * sb * cg + sa * sg2;
          R[3] = sa * cb; R[4] = sa * sb * sg2 + ca * cg; R[5] = sa * sb * cg - ca * sg2;
          R[6] = -sb;     R[7] = cb * sg2;                R[8] = cb * cg;
          lat[0] = pp[3];
        } else { lat[0] = pp[1]; lat[1] = pp[3]; }      // ball_lat: th_p, r_p
        const float* cq = W(A.L.cphq), *cv = W(A.L.cphv);
        for (int j = 0; j < D / 2; ++j) {               // phase in revolutions (the kernels' sin/cos take 2 pi t)
          o[enf_lt_off_phq(H, D) + j] = lat[0] * cq[j] + lat[1] * cq[D / 2 + j];
          o[enf_lt_off_phv(H, D) + j] = lat[0] * cv[j] + lat[1] * cv[D / 2 + j];
        }
      }
    }
  }
}

#ifndef ENF_PROLOGUE_MFMA     // bit 0: K1, bit 1: its backward, as matrix-pipe kernels over 16 latents per workgroup (enf_prologue.hip);
#define ENF_PROLOGUE_MFMA 2   // clear: the kernels here.  Round 3, same box: K1 20.0 us here / 21.0 there, backward 32.0 here / 25.5 there
#endif
extern "C" int enf_launch_prologue_mfma(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*, float*,
                                        float*, float*, hipStream_t);
extern "C" int enf_launch_prologue_bwd_mfma(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*,
                                            const float*, const float*, float*, float*, float*, float*, hipStream_t);

extern "C" int enf_launch_prologue(const EnfDims& m, const EnfLayout& L, const char* blob, const float* p, const float* a,
                                   const float* sigma, float* lt, float* an, float* kv, hipStream_t st) {
#if ENF_PROLOGUE_MFMA & 1
  if (m.D % 32 == 0) return enf_launch_prologue_mfma(m, L, blob, p, a, sigma, lt, an, kv, st);
#endif
  PrologueArgs A;
  A.p = p; A.a = a; A.sigma = sigma; A.blob = blob; A.L = L; A.lt = lt; A.an = an; A.kv = kv;
  A.BZ = m.B * m.Z; A.H = m.H; A.D = m.D; A.C = m.C; A.dp = m.dp; A.inv = m.inv; A.Dt = m.Dt;
  const size_t smem = sizeof(float) * (ZT * m.C + ZT * m.D + ZT * m.HD);
  hipLaunchKernelGGL(enf_prologue_kernel, dim3((A.BZ + ZT - 1) / ZT), dim3(256), smem, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}

// ---------------------------------------------------------------- K1 backward
// dlt (B*Z rows: du | dv0 | dc | dpose | dwcoef) -> dp, da, dsigma.  Same scheme as the forward: gradient rows
// staged in LDS as [feature][latent], the reduction index off the lanes (mu as stored is [h][i][d], i.e. the
// output d of the transposed product is contiguous; a_to_k / a_to_v use their transposed copies wkT / wvT).
struct PrologueBwdArgs {
  const float* p; const float* sigma; const char* blob; EnfLayout L;
  const float* an; const float* kv; const float* dlt;
  float* dp; float* da; float* dsigma;
  int BZ, H, D, C, dp_dim, inv;
  int Dt;
  float* pg;     // weight-gradient backward (or NULL): per latent row [d k (HD) | d an (D) | d s (D) | an (D) | d an * xn (D) | d c_h k_h (HD)],
                 // the operands of the prologue's X^T delta products and column sums (enf_train.hip); an = the LayerNorm's affine output
};

__global__ __launch_bounds__(256) void enf_prologue_bwd_kernel(PrologueBwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = A.D, H = A.H, HD = H * D, C = A.C;
  float* s_du = sm;                    // [HD][ZT]  d u
  float* s_dk = s_du + ZT * HD;        // [2HD][ZT] d k | d v0
  float* s_dan = s_dk + ZT * 2 * HD;   // [D][ZT]
  float* s_dc = s_dan + ZT * D;        // [H][ZT]
  float* s_part = s_dc + ZT * H;       // [256][ZT] partial sums of the d(an) stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row0 = blockIdx.x * ZT;
  const int stride = enf_lt_stride(H, D);
  auto W = [&](size_t off) { return reinterpret_cast<const float*>(A.blob + off); };
  for (int t = tid; t < ZT * HD; t += 256) {
    const int zz = t / HD, j = t % HD, r = row0 + zz;
    const float* g = A.dlt + (size_t)(r < A.BZ ? r : 0) * stride;
    s_du[j * ZT + zz] = r < A.BZ ? g[enf_lt_off_u(H, D) + j] : 0.f;
    s_dk[(HD + j) * ZT + zz] = r < A.BZ ? g[enf_lt_off_v0(H, D) + j] : 0.f;
  }
  if (tid < ZT * H) {
    const int zz = tid / H, h = tid % H, r = row0 + zz;
    s_dc[h * ZT + zz] = r < A.BZ ? A.dlt[(size_t)r * stride + enf_lt_off_c(H, D) + h] : 0.f;
  }
  __syncthreads();
  // dk[h*D+d] = sum_i MU_h[i][d] du_h[i] + cvec_h[d] dc_h
  for (int j = tid; j < HD; j += 256) {
    const int h = j / D, dd = j % D;
    const float cv = W(A.L.cvec)[h * D + dd];
    const float* mu = W(A.L.mu) + (size_t)h * D * D + dd;
    float acc[ZT];
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) acc[zz] = cv * s_dc[h * ZT + zz];
#pragma unroll PRO_UNROLL
    for (int i = 0; i < D; ++i) {
      const float w = mu[(size_t)i * D];
      const float* gp = s_du + (h * D + i) * ZT;
#pragma unroll
      for (int q = 0; q < ZT / 4; ++q) {
        const pf4 x = *reinterpret_cast<const pf4*>(gp + 4 * q);
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[4 * q + zz] = fmaf(x[zz], w, acc[4 * q + zz]);
      }
    }
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) {
      s_dk[j * ZT + zz] = acc[zz];
      if (A.pg && row0 + zz < A.BZ) {
        float* pr = A.pg + (size_t)(row0 + zz) * (2 * HD + 4 * D);
        pr[j] = acc[zz];
        pr[HD + 4 * D + j] = s_dc[h * ZT + zz] * A.kv[(size_t)(row0 + zz) * 2 * HD + j];
      }
    }
  }
  __syncthreads();
  // d(an_affine)[d] = sum_j Wk[d][j] dk[j] + Wv[d][j] dv0[j]  (wkT/wvT: [j][d]); then through scale: dxn = dy * g.
  // D outputs only: the 256 / D thread groups each take a slice of j and the partial sums meet in LDS
  {
    const int nsplit = 256 / D, d = tid % D, part = tid / D, jper = HD / nsplit;
    float acc[ZT];
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) acc[zz] = 0.f;
    const float* wkt = W(A.L.wkt) + d;
    const float* wvt = W(A.L.wvt) + d;
#pragma unroll PRO_UNROLL
    for (int j = part * jper; j < (part + 1) * jper; ++j) {
      const float wk = wkt[(size_t)j * D], wv = wvt[(size_t)j * D];
#pragma unroll
      for (int q = 0; q < ZT / 4; ++q) {
        const pf4 kq = *reinterpret_cast<const pf4*>(s_dk + j * ZT + 4 * q), vq = *reinterpret_cast<const pf4*>(s_dk + (HD + j) * ZT + 4 * q);
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[4 * q + zz] = fmaf(wk, kq[zz], fmaf(wv, vq[zz], acc[4 * q + zz]));
      }
    }
#pragma unroll
    for (int zz = 0; zz < ZT; ++zz) s_part[tid * ZT + zz] = acc[zz];
    __syncthreads();
    if (tid < D) {
      const float g = W(A.L.lna_g)[tid];
#pragma unroll
      for (int zz = 0; zz < ZT; ++zz) {
        float v = 0.f;
        for (int q = 0; q < nsplit; ++q) v += s_part[(q * D + tid) * ZT + zz];
        s_dan[tid * ZT + zz] = v * g;
        if (A.pg && row0 + zz < A.BZ) {
          float* pr = A.pg + (size_t)(row0 + zz) * (2 * HD + 4 * D) + HD;
          const float xn = A.an[(size_t)(row0 + zz) * (2 * D + 2) + D + tid];
          pr[tid] = v;
          pr[2 * D + tid] = xn * g + W(A.L.lna_b)[tid];
          pr[3 * D + tid] = v * xn;
        }
      }
    }
  }
  __syncthreads();
  // LayerNorm backward: ds = rstd * (dxn - mean(dxn) - xn * mean(dxn * xn)); one wave per row
  for (int zz = wave; zz < ZT; zz += 4) {
    const int r = row0 + zz;
    const int rr = r < A.BZ ? r : A.BZ - 1;
    const float* anr = A.an + (size_t)rr * (2 * D + 2);
    float v1 = 0.f, v2 = 0.f;
    for (int d = lane; d < D; d += 64) { const float g = s_dan[d * ZT + zz]; v1 += g; v2 += g * anr[D + d]; }
    const float m1 = wave_sum(v1) / A.Dt, m2 = wave_sum(v2) / A.Dt;
    const float rstd = anr[2 * D + 1];
    for (int d = lane; d < D; d += 64) {
      const float ds = rstd * (s_dan[d * ZT + zz] - m1 - anr[D + d] * m2);
      s_dan[d * ZT + zz] = ds;
      if (A.pg && r < A.BZ) A.pg[(size_t)r * (2 * HD + 4 * D) + HD + D + d] = ds;
    }
  }
  __syncthreads();
  // da[c] = sum_d Ws[c][d] ds[d]
  for (int t = tid; t < ZT * C; t += 256) {
    const int zz = t / C, c = t % C, r = row0 + zz;
    float sacc = 0.f;
#pragma unroll PRO_UNROLL
    for (int d = 0; d < D; ++d) sacc = fmaf(W(A.L.stem_w)[(size_t)c * D + d], s_dan[d * ZT + zz], sacc);
    if (r < A.BZ) A.da[(size_t)r * C + c] = sacc;
  }
  if (tid < ZT) {
    const int r = row0 + tid;
    if (r < A.BZ) {
      const float* g = A.dlt + (size_t)r * stride + enf_lt_off_pose(H, D);
      const float* pp = A.p + (size_t)r * A.dp_dim;
      float* o = A.dp + (size_t)r * A.dp_dim;
      const bool sph = A.inv == ENF_INV_LATITUDE_PERIODIC || A.inv == ENF_INV_POLAR_PERIODIC || enf_inv_has_phase(A.inv);
      if (A.inv == ENF_INV_PONITA || A.inv == ENF_INV_PONITA_FULL) { o[0] = g[0]; o[1] = g[1]; o[2] = -sinf(pp[2]) * g[2] + cosf(pp[2]) * g[3]; }
      else if (sph) { o[0] = g[0]; o[1] = g[1] + cosf(pp[1]) * g[2] - sinf(pp[1]) * g[3]; }
      else { for (int i = 0; i < A.dp_dim && i < 3; ++i) o[i] = g[i]; }
      if (enf_inv_has_phase(A.inv)) {
        const float* e = A.dlt + (size_t)r * stride + enf_lt_off_ext(H, D);     // d R (9) | d(latent-only invariants) (2)
        if (A.inv == ENF_INV_BALL) {
          const float ca = cosf(pp[0]), sa = sinf(pp[0]), cb = cosf(pp[1]), sb = sinf(pp[1]), cg = cosf(pp[2]), sg2 = sinf(pp[2]);
          // chain rule through R(alpha, beta, gamma)
          o[0] += e[0] * (-sa * cb) + e[1] * (-sa * sb * sg2 - ca * cg) + e[2] * (-sa * sb * cg + ca * sg2) +
                  e[3] * (ca * cb) + e[4] * (ca * sb * sg2 - sa * cg) + e[5] * (ca * sb * cg + sa * sg2);
          o[1] += e[0] * (-ca * sb) + e[1] * (ca * cb * sg2) + e[2] * (ca * cb * cg) + e[3] * (-sa * sb) + e[4] * (sa * cb * sg2) +
                  e[5] * (sa * cb * cg) + e[6] * (-cb) + e[7] * (-sb * sg2) + e[8] * (-sb * cg);
          o[2] = e[1] * (ca * sb * cg + sa * sg2) + e[2] * (-ca * sb * sg2 + sa * cg) + e[4] * (sa * sb * cg - ca * sg2) +
                 e[5] * (-sa * sb * sg2 - ca * cg) + e[7] * (cb * cg) + e[8] * (-cb * sg2);
          o[3] = e[9];
        } else { o[1] += e[9]; o[2] = 0.f; o[3] = e[10]; }
      }
      const float sg = A.sigma ? A.sigma[r] : 1.f;
      const float dwc = A.dlt[(size_t)r * stride + enf_lt_off_wcoef(H, D)];
      A.dsigma[r] = (sph ? -1.f : -2.f) / (sg * sg * sg) * dwc;
    }
  }
}

extern "C" int enf_launch_prologue_bwd_wg(const EnfDims& m, const EnfLayout& L, const char* blob, const float* p,
                                          const float* sigma, const float* an, const float* kv, const float* dlt, float* dp,
                                          float* da, float* dsigma, float* pg, hipStream_t st);
extern "C" int enf_launch_prologue_bwd(const EnfDims& m, const EnfLayout& L, const char* blob, const float* p,
                                       const float* sigma, const float* an, const float* kv, const float* dlt, float* dp,
                                       float* da, float* dsigma, hipStream_t st) {
  return enf_launch_prologue_bwd_wg(m, L, blob, p, sigma, an, kv, dlt, dp, da, dsigma, nullptr, st);
}
extern "C" int enf_launch_prologue_bwd_wg(const EnfDims& m, const EnfLayout& L, const char* blob, const float* p,
                                          const float* sigma, const float* an, const float* kv, const float* dlt, float* dp,
                                          float* da, float* dsigma, float* pg, hipStream_t st) {
#if ENF_PROLOGUE_MFMA & 2
  if (m.D % 32 == 0) return enf_launch_prologue_bwd_mfma(m, L, blob, p, sigma, an, kv, dlt, dp, da, dsigma, pg, st);
#endif
  PrologueBwdArgs A;
  A.pg = pg;
  A.p = p; A.sigma = sigma; A.blob = blob; A.L = L; A.an = an; A.kv = kv; A.dlt = dlt;
  A.dp = dp; A.da = da; A.dsigma = dsigma;
  A.BZ = m.B * m.Z; A.H = m.H; A.D = m.D; A.C = m.C; A.dp_dim = m.dp; A.inv = m.inv; A.Dt = m.Dt;
  const size_t smem = sizeof(float) * (ZT * 3 * m.HD + ZT * m.D + ZT * m.H + ZT * 256);
  hipLaunchKernelGGL(enf_prologue_bwd_kernel, dim3((A.BZ + ZT - 1) / ZT), dim3(256), smem, st, A);
  return hipGetLastError() == hipSuccess ? 0 : ENF_ELAUNCH;
}
