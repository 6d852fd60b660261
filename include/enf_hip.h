/* enf_hip.h -- C-ABI of the MI355X (gfx950) Equivariant-Neural-Field decoder.
 *
 * The reference (david-knigge/enf-pde) exposes this path as a Flax module, not an FFI:
 *   nef.apply(params, x, p, a, gaussian_window)            enf/models/equivariant_cross_attention_nef.py:204-235
 *   jax.grad(loss)(latents)   (MAML inner step)            experiments/fitting/trainers/pde_trainer.py:175-200
 * The entry points below are what a binding for that module would bind: one
 * forward, one backward-to-latents, weight packing, and size queries.  Plain pointers and
 * sizes only; every buffer is owned by the caller and lives in device (HBM) memory; nothing
 * is allocated, freed or synchronised inside the library; all work is enqueued on `stream`
 * (a hipStream_t passed as void*; the calling thread's current device must be the stream's).  Functions return 0 or a
 * negative ENF_E* code.  The library keeps no settings: a call depends on its arguments only, calls on different
 * (workspace, stream) pairs are independent and may come from different host threads, models and devices.  A workspace
 * is scratch of ONE call sequence at a time (the REUSE flags below tie a backward to the forward before it).
 *
 * Layouts (all fp32, C-contiguous, batch first):
 *   x      (B, N, dx)   query coordinates; x_bstride = element stride between signals
 *                       (0 = one grid broadcast over B, pde_trainer.py:197,393)
 *   p      (B, Z, dp)   latent poses  (dp = z_pos + z_ori; ponita carries the raw angle,
 *                       the cos/sin embed of NEF:214-217 happens inside)
 *   a      (B, Z, C)    latent context codes
 *   sigma  (B, Z, 1)    gaussian window size per latent
 *   out    (B, N, O)
 */
#ifndef ENF_HIP_H
#define ENF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENF_ABI_VERSION 2

/* cross-attention invariants: enf/steerable_attention/invariant/__init__.py:47-78 */
enum {
  ENF_INV_REL_POS_PERIODIC = 0,  /* rel_pos_periodic.py:35-60,  window _base_invariant.py:35-43 */
  ENF_INV_LATITUDE_PERIODIC = 1, /* spherical_longitude.py:57-85, window :34-55                  */
  ENF_INV_POLAR_PERIODIC = 2,    /* polar_periodic.py:40-68,    window :35-38                    */
  ENF_INV_PONITA = 3,            /* ponita.py:20-44 (PonitaPos2D), window _base_invariant.py:25-33 */
  ENF_INV_ABS_POS = 4,           /* abs_pos.py:42                                                */
  ENF_INV_REL_POS = 5,           /* rel_pos.py:41                                                */
  ENF_INV_NORM_REL_POS = 6,      /* norm_rel_pos.py:34                                           */
  ENF_INV_BALL = 7,              /* ball.py:54-96: [R(alpha,beta,gamma) x^, r_x, r_p], window :36-52 (64-wide kernels only) */
  ENF_INV_BALL_LAT = 8,          /* ball_lat.py:54-88: [th_x, th_p, cos dphi, sin dphi, r_x, r_p]                              */
  ENF_INV_PONITA_FULL = 9,       /* ponita.py:48-92 (Ponita2D): both sides carry an orientation, x = (pos_x, pos_y, theta_x), dx = 3:
                                    [ponita's two, cos(theta_x - theta_p)]; the self-attention invariant of `ponita` (INV/__init__.py:30-32) */
  ENF_INV_COUNT = 10
};

/* arithmetic of the per-pair contractions */
enum {
  ENF_PREC_F32 = 0,  /* v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate (parity mode) */
  ENF_PREC_BF16 = 1  /* v_mfma_f32_16x16x32_bf16: bf16 operands, fp32 accumulate (throughput mode) */
};

/* pair-kernel variant of a call: latent-split / unfolded, or z-fold (per-latent folded matrices) */
enum {
  ENF_VARIANT_AUTO = 0,          /* by problem size */
  ENF_VARIANT_LATENT_SPLIT = 1,  /* forward: the 8 waves split Z; backward: one wave = one latent */
  ENF_VARIANT_ZFOLD = 2,         /* forward: >= 192 workgroups of 128 queries; backward: >= 192 latents */
  ENF_VARIANT_ZFOLD_ZSPLIT = 3   /* forward only: the z-fold kernel over equal runs of latent steps -- the (signal, 128-query tile, latent) space
                                    cut into <= 256 runs, one workgroup each, a tile's partial softmax sums merged by a small kernel
                                    (enf_pair_partition).  AUTO picks it when fewer than 192 tiles exist and Z >= 128 -- e.g. 128 latents on
                                    a 96 x 48 sphere grid, 4 signals: 144 tiles -> 256 runs of 72.  In this variant the last bits of a
                                    query's value depend on the call's shape (the run boundaries order the partial sums) */
};

/* relu masks of a call (see "Relu masks" below) */
#define ENF_MASK_OFF 0
#define ENF_MASK_WRITE 1
#define ENF_MASK_READ 2

enum {
  ENF_OK = 0,
  ENF_EINVAL = -1,       /* NULL pointer / non-positive size */
  ENF_EINVARIANT = -2,   /* unknown invariant id (reference: ValueError, invariant/__init__.py:78) */
  ENF_EUNSUPPORTED = -3, /* shape outside the compiled kernel set (D in {64,128}; H in {1,2}, 4 at D = 64; O <= 32) */
  ENF_EWORKSPACE = -4,   /* workspace too small */
  ENF_ELAUNCH = -5,      /* HIP launch error */
  ENF_EDIM = -6          /* dx / dp inconsistent with the invariant (reference asserts, :62,65) */
};

typedef struct EnfDesc {
  int32_t B, N, Z;       /* signals, queries per signal, latents per signal */
  int32_t H, D, C, O;    /* num_heads, num_hidden, latent_dim, num_out (NEF:85-89) */
  int32_t dx;            /* coordinate width (cfg.nef.num_in) */
  int32_t invariant_id;  /* ENF_INV_* */
  int32_t use_window;    /* use_gaussian_window (NEF:96) */
  int32_t precision;     /* ENF_PREC_* */
  int32_t h_true;       /* 0, or the model's num_heads when H is padded with all-zero heads (num_heads = 3 runs as H = 4):
                           only the block FFN's LayerNorm, which normalises over num_heads * num_hidden, needs it */
  int32_t d_true;       /* 0, or the model's num_hidden when D is a zero-padded width (d_true < D): LayerNorm statistics and
                           the D^-1/2 logit scale use d_true; the caller pads every weight tensor with zeros (see
                           enf-pde_amd/enf/models/_pad.py).  Lets num_hidden 16 / 32 (config_diff_sphere.yaml) run on the D = 64 kernels */
  /* ---- per-call options.  Everything a call depends on is in its arguments: the library keeps no mutable settings. */
  int32_t pair_fwd_variant; /* ENF_VARIANT_*: forward pair kernel.  enf_workspace_bytes / enf_pair_scratch_bytes depend on it */
  int32_t pair_bwd_variant; /* ENF_VARIANT_*: backward pair kernel (the weight-gradient path always runs the unfolded one) */
  int32_t mask_mode;        /* ENF_MASK_*: what the pair kernels of THIS call do with `relu_masks` */
  int32_t mask_signals;     /* signals b, b + mask_signals, ... share the masks of signal b % mask_signals (0 = B) */
  int32_t reserved;
  void* relu_masks;         /* enf_relu_mask_bytes(d) bytes of device memory, or NULL with ENF_MASK_OFF */
} EnfDesc;

/* Weight tensors in the order `enf_pack_weights` expects them; names are the Flax tree of
 * EquivariantCrossAttentionNeF (SURVEY.md 8a "Weights"); kernels are (in, out), y = x@W + b. */
enum {
  ENF_W_STEM_W = 0, ENF_W_STEM_B,                 /* latent_stem                       NEF:134 */
  ENF_W_LNA_G, ENF_W_LNA_B,                       /* cross_attention_blocks_0/layer_norm_attn NEF:29 */
  ENF_W_RQ_COEF, ENF_W_RQ_W1, ENF_W_RQ_B1, ENF_W_RQ_W2, ENF_W_RQ_B2,  /* attn/invariant_embedding_query RFF:21-40 */
  ENF_W_RV_COEF, ENF_W_RV_W1, ENF_W_RV_B1, ENF_W_RV_W2, ENF_W_RV_B2,  /* attn/invariant_embedding_value */
  ENF_W_Q_W, ENF_W_Q_B,                           /* attn/inv_emb_to_q                 ECA:54 */
  ENF_W_K_W, ENF_W_K_B,                           /* attn/a_to_k                       ECA:55 */
  ENF_W_V_W, ENF_W_V_B,                           /* attn/a_to_v                       ECA:56 */
  ENF_W_F1_W0, ENF_W_F1_B0, ENF_W_F1_G, ENF_W_F1_BE, ENF_W_F1_W1, ENF_W_F1_B1,  /* attn/inv_emb_to_v  ECA:65 */
  ENF_W_MX_W0, ENF_W_MX_B0, ENF_W_MX_G, ENF_W_MX_BE, ENF_W_MX_W1, ENF_W_MX_B1,  /* attn/inv_emb_cond_mixer ECA:66 */
  ENF_W_AO_W, ENF_W_AO_B,                         /* attn/out_proj                     ECA:72 */
  ENF_W_FF_W0, ENF_W_FF_B0, ENF_W_FF_G, ENF_W_FF_BE, ENF_W_FF_W1, ENF_W_FF_B1,  /* pointwise_ffn     NEF:40-42 */
  ENF_W_O0_W, ENF_W_O0_B, ENF_W_O2_W, ENF_W_O2_B, ENF_W_O4_W, ENF_W_O4_B,       /* out_proj layers_0/2/4 NEF:196-202 */
  ENF_NUM_TENSORS
};

int enf_abi_version(void);
const char* enf_strerror(int code);

/* invariant metadata (mirrors BaseInvariant.dim / num_z_pos_dims / num_z_ori_dims, _base_invariant.py:7-23) */
int enf_invariant_dim(int invariant_id, int dx);        /* I, or ENF_EINVARIANT */
int enf_invariant_pose_dim(int invariant_id, int dx);   /* dp = z_pos + z_ori    */

/* validate a descriptor against the compiled kernel set */
int enf_check_desc(const EnfDesc* d);

/* bytes of the packed weight blob (MFMA-fragment-ordered panels + folded matrices) */
size_t enf_packed_weight_bytes(const EnfDesc* d);
/* Build the packed blob from ENF_NUM_TENSORS device pointers (fp32).  Runs the exact
 * algebraic folds of DESIGN.md ("folds") in fp32 on the device.  `tensors` is a host array. */
int enf_pack_weights(const EnfDesc* d, const float* const* tensors, void* packed, void* stream);

/* scratch the forward / backward need (latent table, per-query partials) */
size_t enf_workspace_bytes(const EnfDesc* d);

/* Replaces nef.apply (NEF:204-235).  `ybar` (B,N,H*D) and `lse` (B,N,H) are optional
 * outputs (may be NULL): the attention-weighted value sum and the softmax log-sum-exp,
 * which enf_backward_latents takes back instead of recomputing the forward. */
int enf_forward(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a,
                const float* sigma, const void* packed, float* out, float* ybar, float* lse,
                void* workspace, size_t workspace_bytes, void* stream);

/* Measurement hook: the same call as enf_forward, restricted to a subset of its kernels so that ONE
 * kernel can be bracketed by events on `stream` (bench.py's roofline leg, rocprofv3 cross-check).
 * stages: bit 0 = latent prologue (K1), bit 1 = pair kernel (K2), bit 2 = tail.  The latent table
 * lives in `workspace` between calls, so run bit 0 once before timing bit 1. */
#define ENF_STAGE_PROLOGUE 1u
#define ENF_STAGE_PAIR 2u
#define ENF_STAGE_TAIL 4u
#define ENF_STAGE_FOLD 8u      /* enf_wz_kernel: per-latent folded matrices of the z-fold pair variant (no-op otherwise);
                                  runs between PROLOGUE and PAIR, PAIR alone reuses what the workspace holds */
#define ENF_STAGE_TAIL_SAVE 16u /* with ENF_STAGE_TAIL: stash the tail's pre-activations in the workspace; the backward
                                  that follows on the untouched workspace (ENF_BWD_REUSE_TAIL) then skips their recompute */
#define ENF_STAGE_PREPARE_BWD 32u /* a backward on the same inputs and WORKSPACE follows: what it needs from the latent table
                                  alone (its per-latent folded matrices, the zeroed gradient table) starts on the device's
                                  side stream behind the pair kernel; pass ENF_BWD_REUSE_PREPARED to that backward.  The
                                  pending work is recorded against this workspace: any later call on the same workspace
                                  joins it first, a call on another workspace neither sees nor consumes it */
#define ENF_STAGE_YBAR_HALF 64u /* with ENF_STAGE_PAIR | ENF_STAGE_TAIL in ONE call, bf16 mode, `ybar` == NULL: nothing after this call reads
                                   its `ybar` (a decode: no backward follows), so the pair kernel may hand it to the tail as bf16 in the
                                   workspace -- half the bytes of the largest tensor of a forward; the tail rounds `ybar` to bf16 for its
                                   first matrix product anyway, so `out` is the same bit for bit.  Ignored where it does not apply
                                   (f32 mode, ENF_STAGE_TAIL_SAVE, the split z-fold variant, a caller-owned `ybar`). */
int enf_forward_stages(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a,
                       const float* sigma, const void* packed, float* out, float* ybar, float* lse,
                       void* workspace, size_t workspace_bytes, unsigned stages, void* stream);

/* Replaces jax.grad(loss)(latents) for one inner step (pde_trainer.py:188,200): given
 * dL/dout it returns dL/dp (B,Z,dp), dL/da (B,Z,C), dL/dsigma (B,Z,1).  Buffers are
 * overwritten, not accumulated.  `ybar`/`lse` come from enf_forward on the same inputs. */
int enf_backward_latents(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p,
                         const float* a, const float* sigma, const void* packed, const float* ybar,
                         const float* lse, const float* dout, float* dp, float* da, float* dsigma,
                         void* workspace, size_t workspace_bytes, void* stream);
/* flags: ENF_BWD_REUSE_PROLOGUE = the workspace still holds the latent table of the enf_forward call with the same
 * (p, a, sigma, weights) -- nothing else has used it since -- so the prologue is not recomputed. */
#define ENF_BWD_REUSE_PROLOGUE 1u
/* ENF_BWD_REUSE_TAIL (with ENF_BWD_REUSE_PROLOGUE): that forward also ran with ENF_STAGE_TAIL_SAVE: the tail backward
 * reads the stashed pre-activations instead of recomputing the tail's forward chain (half of its GEMM stages). */
#define ENF_BWD_REUSE_TAIL 2u
/* ENF_BWD_REUSE_PREPARED (with ENF_BWD_REUSE_PROLOGUE): that forward ran with ENF_STAGE_PREPARE_BWD: join its side-stream
 * work instead of repeating it (ignored when nothing is pending). */
#define ENF_BWD_REUSE_PREPARED 4u
/* ENF_BWD_ONLY_PAIR: measurement hook, the backward counterpart of enf_forward_stages(ENF_STAGE_PAIR): re-run ONLY the
 * backward pair kernel (preceded by the zero-fill of the 2 MB gradient table it accumulates into) on the workspace a
 * complete enf_backward_latents[_ex] call with the same arguments has just left; dp / da / dsigma are not written. */
#define ENF_BWD_ONLY_PAIR 8u
int enf_backward_latents_ex(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p,
                         const float* a, const float* sigma, const void* packed, const float* ybar,
                         const float* lse, const float* dout, float* dp, float* da, float* dsigma,
                         void* workspace, size_t workspace_bytes, unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training path: gradients w.r.t. the network weights (value_and_grad over params['nef'],
 * pde_trainer.py:255; nonmaml_pde_trainer.py:304-339).  The per-pair chain (97 % of the FLOPs)
 * stays in HIP -- enf_pair_forward for the values, enf_backward_weights for d lt and the gradients of the ENF_P_* tensors --;
 * folds, latent prologue and tail (per-latent / per-query work) are run by the host framework as differentiable ops around
 * them.
 *   latent table `lt` (B*Z rows, enf_lt_layout): [ u (H*D) | v0 (H*D) | pose (4) | wcoef | pad | c (H) | pad ]
 *     u, c   : att[n,z,h] = h1[n,z,:].u[z,h,:] + c[z,h]        (fold 1)
 *     v0     : a_to_v(a_norm)                                   (ECA:94)
 *     pose   : periodic/rel/abs/norm (p0,p1,p2,-); ponita (px,py,cos t,sin t); sphere (phi,theta,sin theta,cos theta)
 *     wcoef  : 1/sigma^2 (sphere: 1/(2 sigma^2))
 * ------------------------------------------------------------------------------------------- */
enum {   /* effective per-pair parameters, plain fp32, kernels (in, out) */
  ENF_P_AQ1 = 0, ENF_P_BQ1,   /* query RFFNet layers_0                         (D,D), (D)  */
  ENF_P_AV1, ENF_P_BV1,       /* value RFFNet layers_0                         (D,D), (D)  */
  ENF_P_AF, ENF_P_BF,         /* linear_final . inv_emb_to_v.Dense_0 (fold 2)  (D,D), (D)  */
  ENF_P_AGB, ENF_P_BGB,       /* diag(LN.scale) Dense_1, LN.bias Dense_1 + b; Flax column order [gamma (HD) | beta (HD)]  (D,2HD), (2HD) */
  ENF_P_AM, ENF_P_BM,         /* inv_emb_cond_mixer.Dense_0                    (D,D), (D)  */
  ENF_P_COEFQ, ENF_P_COEFV,   /* RFF coefficients                              (I, D/2)    */
  ENF_NUM_PAIR_TENSORS
};
/* Per-pair activations / deltas the backward can materialise (rows = (b*Z + z)*N + n, D columns,
 * bf16 in ENF_PREC_BF16 and fp32 in ENF_PREC_F32) so that every per-pair weight gradient is a plain
 * GEMM dW = X^T delta over the pair axis:   AQ1: EQ^T DA1   AV1: EV^T DA2   AF: G1^T DA3
 *   AGB[:, gamma_h|beta_h]: NH^T DG_h | NH^T DB_h     AM: sum_h V_h^T DA5_h    biases: column sums of delta */
enum { ENF_S_EQ = 0, ENF_S_EV, ENF_S_G1, ENF_S_NH, ENF_S_DA1, ENF_S_DA2, ENF_S_DA3, ENF_S_HEAD0 /* + 4h: V, DA5, DG, DB */ };
/* Column order of the stored rows.  ENF_PREC_F32: natural.  ENF_PREC_BF16: permuted inside every block of 32 columns --
 * stored column 32 b + 8 q + j (q < 4, j < 8) holds feature 32 b + 4 q + j for j < 4 and 32 b + 16 + 4 q + (j - 4)
 * otherwise (the MFMA operand fragments go out as they sit in registers, one 16-byte store each).  All buffers share the
 * permutation, so X^T delta comes out with rows and columns permuted alike: un-permute the D x D result. */
#define ENF_NUM_STORE(H) (7 + 4 * (H))

int enf_lt_layout(const EnfDesc* d, int* stride, int* off_u, int* off_v0, int* off_pose, int* off_wcoef, int* off_c);
/* ball / ball_lat only (invariant/ball.py, ball_lat.py): further fields of a latent-table row.
 *   ext (16 floats): [ R (9, row-major; ball.py:76-84) | the latent-only invariants (2) | pad ]; in the GRADIENT table the
 *                    backward returns d R and d(latent-only invariants) in the same slots
 *   phase_q, phase_v (D/2 each): coeff[latent-only rows]^T (latent-only invariants), the per-latent part of the RFF
 *                    pre-activation of the query / value RFFNet */
int enf_lt_layout_ext(const EnfDesc* d, int* off_ext, int* off_phase_q, int* off_phase_v);
int enf_pack_pair(const EnfDesc* d, const float* const* pair_tensors, void* packed, void* stream);
/* K2 alone: lt -> ybar (B,N,H*D), lse (B,N,H).  `scratch` (enf_pair_scratch_bytes; may be 0 -> NULL) holds the
 * per-latent folded matrices of the large-N forward variant. */
size_t enf_pair_scratch_bytes(const EnfDesc* d);
int enf_pair_forward(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                     float* ybar, float* lse, void* scratch, size_t scratch_bytes, void* stream);
/* K3 alone: d ybar, delta[n,h] = d ybar . ybar, lse -> d lt (same layout as lt, overwritten);
 * `store` = NULL or ENF_NUM_STORE(H) device buffers of B*Z*N rows (see ENF_S_*). */
int enf_pair_backward(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                      const float* lse, const float* dybar, const float* delta, float* dlt, void* const* store,
                      void* stream);
/* The same with the gradient w.r.t. the QUERY coordinates (jax.grad of nef.apply w.r.t. x; the self-attention blocks of
 * NEF:223-226 need it, their queries being the latent poses): `dx` (B,N,dx) fp32, ACCUMULATED into with float atomics
 * (zero it first), or NULL. */
int enf_pair_backward_ex(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                         const float* lse, const float* dybar, const float* delta, float* dlt, void* const* store,
                         float* dx, void* stream);

/* Relu masks.  Second-order terms taken as finite differences of FIRST-order gradients (the outer MAML step,
 * pde_trainer.py:255) converge to the DISTRIBUTIONAL second derivative: relu units whose sign changes between the two
 * perturbed points add a finite amount that automatic differentiation (relu'' = 0) never includes (20 % on the RFFNet
 * layer-0 weights in the tests).  With EnfDesc.mask_mode = ENF_MASK_WRITE the pair-kernel forward of the call records, per
 * pair and relu layer (the two RFFNet layers), which pre-activations are positive, into EnfDesc.relu_masks; with
 * ENF_MASK_READ the pair-kernel forward of the call, and enf_pair_backward[_ex] / enf_backward_weights, use the relu
 * LINEARISED at those masks (h = a where the bit is set) instead of max(a, 0), for signals b, b + mask_signals, ...
 * alike.  One 32-bit word per lane, 16-query tile, latent and layer: enf_relu_mask_bytes(d) for the shape that WRITES
 * them (same N, Z; B = mask_signals). */
size_t enf_relu_mask_bytes(const EnfDesc* d);

/* One inner step of the MAML loop (pde_trainer.py:175-207) in ONE call: forward on (x, p, a, sigma), *loss += mean((out - target)^2)
 * (the caller zeroes *loss; target (B,N,O) fp32), and grad_scale * d loss / d(p, a, sigma) into dp / da / dsigma (OVERWRITTEN).
 * Equals enf_forward + enf_mse_value_grad + enf_backward_latents on the same arguments; the per-query tail runs once, as a single
 * kernel (forward chain, loss and its gradient in registers, backward chain), so neither `out` nor `d out` is materialised.
 * `workspace`: enf_workspace_bytes(d), contents need not be kept. */
int enf_fit_step(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a, const float* sigma,
                 const void* packed, const float* target, float grad_scale, float* loss, float* dp, float* da, float* dsigma,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Reconstruction loss of the inner loop and its gradient in one pass (pde_trainer.py:185):
 *   *loss += mean((out - target)^2)   (the caller zeroes *loss),   dout = 2 (out - target) / n * grad_scale  (dout may be NULL) */
int enf_mse_value_grad(const float* out, const float* target, size_t n, float grad_scale, float* dout, float* loss,
                       void* stream);

/* The meta-SGD update of one inner step for all latent components in one launch (pde_trainer.py:206-219):
 *     out = x - lr * (scale * g)        scale = the batch size (:206); lr broadcasts over the leading dims
 * A segment is one component (p_pos, p_ori, a, gaussian_window): x, out contiguous (n elements, trailing dimension
 * `width`); g may be a column slice of a wider array: element (row, c) is g[row * g_stride + c]; lr holds 1 or `width`
 * values (trainers/pde_trainer.py:83-97).  A component whose update is zeroed (:209-212) is simply left out. */
#define ENF_SGD_MAX_SEGMENTS 4
typedef struct EnfSgdSegment {
  const float* x;
  const float* g;
  const float* lr;
  float* out;
  int64_t n;
  int32_t width, g_stride, lr_len, reserved;
} EnfSgdSegment;
int enf_meta_sgd_update(int nseg, const EnfSgdSegment* segs, float scale, void* stream);

/* What the inner loop prepares before its first step, in ONE launch (pde_trainer.py:157-159, 193-197; six framework kernels otherwise):
 *   - every latent component of the shared initialisation, (1, Z, width), repeated for the B signals -> (B, Z, width);
 *   - the coordinates and the targets of all S1 = S + 1 sampled point sets gathered once:
 *       xs[s, i, :] = coords[masks[i, s], :]     (S1, Ns, dx)       ys[s, b, i, :] = img[b, masks[i, s], :]     (S1, B, Ns, O)
 *     with masks (Ns, S1) int64 indices into the N grid points, coords (N, dx), img (B, N, O), all fp32 and contiguous;
 *   - the S1 loss accumulators zeroed. */
typedef struct EnfFitComponent {
  const float* src; /* (1, Z, width) */
  float* dst;       /* (B, Z, width) */
  int32_t width, reserved;
} EnfFitComponent;
int enf_fit_inputs(int ncomp, const EnfFitComponent* comps, int32_t B, int32_t Z, int32_t N, int32_t Ns, int32_t S1, int32_t dx, int32_t O,
                   const float* coords, const float* img, const int64_t* masks, float* xs, float* ys, float* losses, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latent ODE (experiments/fitting/ode_models/ponita_ode_g.py): the separable group convolution of a ConvBlock,
 * SepGconv.__call__ (:63-83), over the fully connected latent set of every signal:
 *     out[b,r,:] = bias + sum_s a[b,s,:] * (kb[b,r,s,:] @ W)        a (B,Z,C), kb (B,Z,Z,J), W (J,C), bias (C) or NULL
 * fused: the (B,Z,Z,C) `kernel` tensor the reference materialises (:72) never exists; the J -> C product runs as fp32
 * MFMA tiles consumed in registers.  kb element (b,r,s,j) is read at b*Z*Z*J + r*kb_stride_r + s*kb_stride_s + j, so
 * the same entry point gives the gradient w.r.t. the senders:  d a = conv(g, kb with the (r,s) strides swapped, W, NULL).
 * enf_ode_conv_backward_basis:  d kb[b,r,s,:] = W (g[b,r,:] * a[b,s,:])   (B,Z,Z,J).
 * enf_ode_conv_backward_weight: d W (J,C) = kb^T (g (x) a) over the pair axis, the (B,Z,Z,C) product formed in registers, and
 *   d bias (C) = sum_{b,r} g: `dW` is ONE buffer of J*C + C floats, d W followed by d bias;
 *   scratch: enf_ode_conv_backward_weight_scratch_bytes(B,Z,J,C) bytes (per-workgroup partials, summed in a fixed order).
 * J, C in {16, 32, 64, 128}; all buffers fp32, contiguous, 16-byte aligned. */
int enf_ode_conv_forward(int B, int Z, int J, int C, const float* a, const float* kb, int64_t kb_stride_r,
                         int64_t kb_stride_s, const float* W, const float* bias, float* out, void* stream);
int enf_ode_conv_backward_basis(int B, int Z, int J, int C, const float* a, const float* g, const float* W, float* dkb,
                                void* stream);
size_t enf_ode_conv_backward_weight_scratch_bytes(int B, int Z, int J, int C);
int enf_ode_conv_backward_weight(int B, int Z, int J, int C, const float* a, const float* kb, const float* g, float* dW,
                                 void* scratch, size_t scratch_bytes, void* stream);
/* PolynomialFeatures (ponita_ode_g.py:15-26) of the P = B Z^2 pair invariants x (P, I): the Kronecker powers
 * [x, x(x)x, ..., x^(x)(degree+1)] concatenated, F = I + I^2 + ... + I^(degree+1) values per pair (enf_ode_poly_num_features;
 * 340 for I = 4, degree = 3), in the reference's order (each power appends its new factor as the last index).
 * forward: feat (P, F);  backward: dx (P, I) = J^T dfeat.  I <= 8, degree <= 7.  No intermediate power is materialised. */
int enf_ode_poly_num_features(int I, int degree);
int enf_ode_poly_forward(int64_t P, int I, int degree, const float* x, float* feat, void* stream);
int enf_ode_poly_backward(int64_t P, int I, int degree, const float* x, const float* dfeat, float* dx, void* stream);
/* The kernel basis of PonitaGen (ponita_ode_g.py:128-131, 158-160), fused:  kb = gelu(gelu(poly(inv) W1 + b1) W3 + b3)
 * over the P = B Z^2 pairs, inv (P, I), W1 (F, H1), b1 (H1), W3 (H1, J), b3 (J) as in the reference's parameter tree
 * (kernel_basis/layers_1, layers_3), kb (P, J); gelu = the tanh form (flax default).  Neither the (P, F) feature tensor nor
 * the (P, H1) hidden layer touch memory (csrc/enf_ode_basis.hip).  backward: given d kb (P, J) -> d inv (P, I), d W1, d b1,
 * d W3, d b3 (all OVERWRITTEN; weight gradients are summed in a fixed order: bitwise reproducible).
 * I in 1..4, degree = 3, H1 in {32, 64, 128} (forward also 256), J in {32, 64, 128} (enf_ode_basis_supported -> 1 / 0);
 * `scratch`: enf_ode_basis_scratch_bytes(P, I, H1, J, backward) bytes, 16-byte aligned, contents need not be kept. */
/* Vector readout of PonitaGen (ponita_ode_g.py:176-193) with one output channel, fused:
 *     out[b,r,:] = mean_s wgt[b,r,s] * (cr * u[b,r,:] + cs * w[b,s,:]),    wgt[b,r,s] = inv[b,r,s,:] . Wi + aw[b,s]
 * inv (B,Z,Z,I) the pair invariants, Wi (I) = the readout kernel's rows for the invariants, aw (B,Z) = a @ (its rows for the latent
 * features), u / w (B,Z,D) receiver / sender vectors (relative position: u = w = p_pos, cr = 1, cs = -1; sender orientation: cr = 0,
 * cs = 1).  backward: g = d out (B,Z,D) -> d inv, d aw, d u, d w (OVERWRITTEN) and dWi_part (B * ceil(Z/64), I): partial sums of
 * d Wi, to be added up by the caller.  I <= 6, D in {2, 3}; no atomics. */
int enf_ode_vec_readout_forward(int B, int Z, int I, int D, const float* inv, const float* aw, const float* u, const float* w,
                                float cr, float cs, const float* Wi, float* out, void* stream);
int enf_ode_vec_readout_backward(int B, int Z, int I, int D, const float* inv, const float* aw, const float* u, const float* w,
                                 float cr, float cs, const float* Wi, const float* g, float* dinv, float* daw, float* du,
                                 float* dw, float* dWi_part, void* stream);
/* The per-latent half of a ConvBlock (ponita_ode_g.py:44-48), fused:  out = Dense_2(gelu(Dense_1(LayerNorm_eps(x))))  over the
 * R = B Z latent rows, x (R, H), gamma / beta (H), W1 (H, M), b1 (M), W2 (M, H), b2 (H) as in the reference's tree
 * (norm/scale, norm/bias, linear_1, linear_2); gelu = the tanh form.  forward also writes pre = LayerNorm(x) W1 + b1 (R, M)
 * for the backward.  backward: d out g (R, H) -> d x (R, H) and, in ONE buffer of 2 H M + M + 3 H floats (OVERWRITTEN),
 * d W1 (H, M) | d W2 (M, H) | d b1 (M) | d b2 (H) | d gamma (H) | d beta (H), summed in a fixed order.
 * H in {32, 64, 128}, M = 2 H (enf_ode_block_supported); scratch: enf_ode_block_scratch_bytes(R, H, M) bytes, 16-byte aligned. */
int enf_ode_block_supported(int H, int M);
size_t enf_ode_block_scratch_bytes(int64_t R, int H, int M);
int enf_ode_block_forward(int64_t R, int H, int M, const float* x, const float* gamma, const float* beta, const float* W1,
                          const float* b1, const float* W2, const float* b2, float eps, float* out, float* pre, void* stream);
int enf_ode_block_backward(int64_t R, int H, int M, const float* x, const float* gamma, const float* beta, const float* W1,
                           const float* W2, const float* pre, const float* g, float eps, float* dx, float* dparams,
                           void* scratch, size_t scratch_bytes, void* stream);
int enf_ode_basis_supported(int I, int degree, int H1, int J, int backward);
size_t enf_ode_basis_scratch_bytes(int64_t P, int I, int H1, int J, int backward);
int enf_ode_basis_forward(int64_t P, int I, int degree, int H1, int J, const float* inv, const float* W1, const float* b1,
                          const float* W3, const float* b3, float* kb, void* scratch, size_t scratch_bytes, void* stream);
int enf_ode_basis_backward(int64_t P, int I, int degree, int H1, int J, const float* inv, const float* W1, const float* b1,
                           const float* W3, const float* b3, const float* dkb, float* dinv, float* dW1, float* db1,
                           float* dW3, float* db3, void* scratch, size_t scratch_bytes, void* stream);

/* EVERY weight gradient in one call (value_and_grad over params['nef'], pde_trainer.py:255; SURVEY.md 8b's
 * enf_backward_weights in full).  Given d out it returns d p, d a, d sigma AND the gradients of all ENF_NUM_TENSORS
 * Flax-named tensors, fp32, shaped like the tensors enf_pack_weights takes, OVERWRITTEN (`dW` = host array of device pointers;
 * the two frozen RFF coefficient entries -- rff.py:87-90 -- may be NULL, else they are zero-filled).  `tensors` are the same
 * weights `packed` was built from (the fold backward needs the unfolded factors), `ybar` / `lse` come from enf_forward[_stages]
 * on the same inputs, `workspace` is that call's workspace (flags: ENF_BWD_REUSE_PROLOGUE / ENF_BWD_REUSE_TAIL as for
 * enf_backward_latents_ex; 0 recomputes what it needs), `dx` (B,N,dx) or NULL accumulates the gradient w.r.t. the queries.
 * Kernels: the tail backward in its weight-gradient form, K3 with the activation store + K4 (enf_xtd_kernel), the prologue
 * backward, fp32 matrix-pipe X^T delta products over the query / latent rows, and the chain rule through the folds of
 * enf_pack_weights -- no library GEMM, no host framework op; slices are summed in a fixed order (same inputs, same bits,
 * up to the float atomics of d lt).  scratch_bytes >= enf_backward_all_scratch_bytes(d, c) for some chunk size c in 1..B
 * signals (the largest c that fits is used; with relu masks a multiple of mask_signals).  EnfDesc.mask_mode = ENF_MASK_READ
 * replays the relu masks in the pair kernel as enf_backward_weights does. */
size_t enf_backward_all_scratch_bytes(const EnfDesc* d, int chunk_signals);
int enf_backward_all(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a, const float* sigma,
                     const float* const* tensors, const void* packed, const float* ybar, const float* lse, const float* dout,
                     float* dp, float* da, float* dsigma, float* const* dW, float* dx, void* workspace, size_t workspace_bytes,
                     void* scratch, size_t scratch_bytes, unsigned flags, void* stream);

/* Weight gradients of the per-pair chain (enf_backward_weights).  enf_pair_backward_ex plus, in the same
 * call, the gradients of the loss w.r.t. the ten trainable ENF_P_* tensors:
 *     dpair[ENF_P_A*] = X^T delta  (in, out),   dpair[ENF_P_B*] = 1^T delta      over all B Z N pairs, fp32, OVERWRITTEN
 * (dpair = host array of ENF_NUM_PAIR_TENSORS device pointers laid out like the tensors enf_pack_pair takes; the two RFF
 * coefficient entries are ignored: frozen in the reference, rff.py:87-90).  Kernels: K3 in its STORE instantiation writes
 * every layer's input / delta fragments for a chunk of signals into `scratch` (ENF_S_* above), enf_xtd_kernel reads each of
 * them ONCE and forms all 3 + 3H products and bias sums on the matrix pipe, a reduction sums slices and chunks in a fixed
 * order (same inputs -> same bits).  No library GEMM, no host framework op.
 *   scratch_bytes >= enf_backward_weights_scratch_bytes(d, c) for some chunk size c in 1..B signals: the call uses the
 *   largest c that fits (with relu masks: a multiple of mask_signals).  c = B: one pass.
 * EnfDesc.mask_mode = ENF_MASK_READ replays the relu masks as in enf_pair_backward_ex with a store. */
size_t enf_backward_weights_scratch_bytes(const EnfDesc* d, int chunk_signals);
int enf_backward_weights(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                         const float* lse, const float* dybar, const float* delta, float* dlt, float* const* dpair,
                         float* dx, void* scratch, size_t scratch_bytes, void* stream);

/* The pair-kernel variant a call with this descriptor runs (ENF_VARIANT_AUTO resolved): ENF_VARIANT_LATENT_SPLIT or
 * ENF_VARIANT_ZFOLD (or, forward only, ENF_VARIANT_ZFOLD_ZSPLIT); `backward` = 0 for the forward kernel, 1 for the backward kernel.
 * Negative ENF_E* on a bad descriptor. */
int enf_pair_variant(const EnfDesc* d, int backward);
/* How ENF_VARIANT_ZFOLD_ZSPLIT cuts the forward pair kernel's work: the flattened (signal, 128-query tile, latent) space of
 * tiles x Z latent steps is walked by `workgroups` workgroups of `run` consecutive steps each (the last one may be shorter); a query
 * tile's latents are met by at most `parts` of them (the partial-sum slots the workspace holds).  Returns 1 and fills the three values when
 * the descriptor resolves to that variant, 0 (values untouched) when it does not, negative ENF_E* on a bad descriptor. */
int enf_pair_partition(const EnfDesc* d, int32_t* run, int32_t* workgroups, int32_t* parts);

#ifdef __cplusplus
}
#endif
#endif /* ENF_HIP_H */
