// enf_api.hip -- the C-ABI of include/enf_hip.h: validation, workspace carving, kernel sequencing.
// No allocation, no host synchronisation, no global state besides one-time kernel attributes.
#include <hip/hip_runtime.h>
#include <mutex>
#include "enf_layout.h"

extern "C" {
int enf_launch_prologue(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*, float*,
                        float*, float*, hipStream_t);
int enf_launch_prologue_bwd(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*,
                            const float*, const float*, float*, float*, float*, hipStream_t);
int enf_launch_pair_fwd(const EnfDims&, const EnfLayout&, const char*, const float*, long long, const float*, float*, float*,
                        char*, float*, char*, int, int, hipStream_t);
int enf_launch_pair_bwd(const EnfDims&, const EnfLayout&, const char*, const float*, long long, const float*, const float*,
                        const float*, const float*, float*, void* const*, const char*, const float*, float*, hipStream_t);
int enf_launch_wz(const EnfDims&, const EnfLayout&, const char*, const float*, char*, float*, char*, char*, hipStream_t);
int enf_launch_tail(const EnfDims&, const EnfLayout&, const char*, const float*, float*, const float*, float*, float*, float*,
                    int, int, hipStream_t);
}

extern "C" int enf_abi_version(void) { return ENF_ABI_VERSION; }

extern "C" const char* enf_strerror(int code) {
  switch (code) {
    case ENF_OK: return "ok";
    case ENF_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case ENF_EINVARIANT: return "Unknown invariant type";
    case ENF_EUNSUPPORTED: return "shape not in the compiled kernel set (num_hidden in {64,128}, num_heads in {1,2}, num_out <= 32)";
    case ENF_EWORKSPACE: return "workspace too small";
    case ENF_ELAUNCH: return "HIP launch failed";
    case ENF_EDIM: return "coordinate / pose width inconsistent with the invariant";
    default: return "unknown error";
  }
}

extern "C" int enf_invariant_dim(int inv, int dx) {
  const int i = enf_inv_dim(inv, dx);
  return i < 0 ? ENF_EINVARIANT : i;
}
extern "C" int enf_invariant_pose_dim(int inv, int dx) {
  const int i = enf_inv_pose_dim(inv, dx);
  return i < 0 ? ENF_EINVARIANT : i;
}

extern "C" int enf_check_desc(const EnfDesc* d) {
  if (!d) return ENF_EINVAL;
  if (d->B <= 0 || d->N <= 0 || d->Z <= 0 || d->C <= 0 || d->O <= 0 || d->H <= 0 || d->D <= 0) return ENF_EINVAL;
  if (d->invariant_id < 0 || d->invariant_id >= ENF_INV_COUNT) return ENF_EINVARIANT;
  if (d->dx < 1 || d->dx > 3) return ENF_EDIM;
  const bool two_d = d->invariant_id == ENF_INV_REL_POS_PERIODIC || d->invariant_id == ENF_INV_PONITA ||
                     d->invariant_id == ENF_INV_LATITUDE_PERIODIC || d->invariant_id == ENF_INV_POLAR_PERIODIC;
  if (two_d && d->dx != 2) return ENF_EDIM;     // reference: assert cfg.num_in == 2 (invariant/__init__.py:62,65)
  if (enf_inv_has_phase(d->invariant_id) && d->D != 64) return ENF_EUNSUPPORTED;   // (the 128-wide backward kernel has no LDS left)
  if (enf_inv_has_phase(d->invariant_id) && d->dx != 3) return ENF_EDIM;   // ball, ball_lat: (phi, theta, r) coordinates
  if (!(d->D == 64 || d->D == 128)) return ENF_EUNSUPPORTED;
  if (d->d_true < 0 || d->d_true > d->D || (d->d_true & 1)) return ENF_EINVAL;
  if (!(d->H == 1 || d->H == 2 || (d->H == 4 && d->D == 64))) return ENF_EUNSUPPORTED;   // 4 heads: 64-wide kernels only
  if (d->h_true < 0 || d->h_true > d->H) return ENF_EINVAL;
  if (d->O > 32) return ENF_EUNSUPPORTED;
  if (d->precision != ENF_PREC_F32 && d->precision != ENF_PREC_BF16) return ENF_EINVAL;
  return ENF_OK;
}

extern "C" size_t enf_packed_weight_bytes(const EnfDesc* d) {
  if (enf_check_desc(d) != ENF_OK) return 0;
  return enf_layout(enf_dims(d)).total;
}

extern "C" size_t enf_workspace_bytes(const EnfDesc* d) {
  if (enf_check_desc(d) != ENF_OK) return 0;
  return enf_workspace(enf_dims(d)).total;
}

// One side stream per process for work that can overlap the caller's stream (created on first use; ENF_SIDE_STREAM=0
// disables it).  Fork / join is by events, so the caller's stream order is preserved; the mutex keeps concurrent host
// threads from interleaving their record / wait pairs on the shared events.
struct SideStream { hipStream_t s; hipEvent_t fork, join; std::mutex mu; bool pending = false; };
static SideStream* side_stream() {
  static SideStream* S = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* e = getenv("ENF_SIDE_STREAM");
    if (!(e && e[0] == '0')) {
      SideStream* t = new SideStream();
      if (hipStreamCreateWithFlags(&t->s, hipStreamNonBlocking) == hipSuccess &&
          hipEventCreateWithFlags(&t->fork, hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&t->join, hipEventDisableTiming) == hipSuccess)
        S = t;
      else
        delete t;
    }
  }
  return S;
}

// join side-stream work left pending by an earlier ENF_STAGE_PREPARE_BWD (no matching backward came) before `st`
// touches the workspace regions it writes
static int side_join_pending(hipStream_t st) {
  SideStream* side = side_stream();
  if (!side) return 0;
  std::lock_guard<std::mutex> lk(side->mu);
  if (side->pending) {
    if (hipStreamWaitEvent(st, side->join, 0) != hipSuccess) return ENF_ELAUNCH;
    side->pending = false;
  }
  return 0;
}

extern "C" int enf_forward_stages(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a,
                                  const float* sigma, const void* packed, float* out, float* ybar, float* lse,
                                  void* workspace, size_t workspace_bytes, unsigned stages, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !p || !a || !packed || !out || !workspace) return ENF_EINVAL;
  if (d->use_window && !sigma) return ENF_EINVAL;
  const EnfDims m = enf_dims(d);
  const EnfLayout L = enf_layout(m);
  const EnfWorkspace W = enf_workspace(m);
  if (workspace_bytes < W.total) return ENF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  const char* blob = (const char*)packed;
  float* yb = ybar ? ybar : F(W.ybar);
  const bool zf = enf_use_zfold(m);
  float* ls = lse ? lse : F(W.lse);
  if ((rc = side_join_pending(st))) return rc;
  if ((stages & ENF_STAGE_PROLOGUE) && (rc = enf_launch_prologue(m, L, blob, p, a, sigma, F(W.lt), F(W.an), F(W.kv), st))) return rc;
  if ((stages & (ENF_STAGE_PAIR | ENF_STAGE_FOLD)) &&
      (rc = enf_launch_pair_fwd(m, L, blob, x, x_bstride, F(W.lt), yb, ls, zf ? ws + W.wz : nullptr, zf ? F(W.wzb) : nullptr,
                                zf ? ws + W.wzu : nullptr, (stages & ENF_STAGE_FOLD) != 0, (stages & ENF_STAGE_PAIR) != 0, st)))
    return rc;
  if ((stages & ENF_STAGE_PREPARE_BWD) && enf_use_zfold_bwd(m)) {
    // what the backward needs from the latent table alone -- its per-latent folded matrices, the zeroed gradient table --
    // starts on the side stream behind the pair kernel, beside the tail, the caller's loss and the tail backward
    SideStream* side = side_stream();
    if (side) {
      std::lock_guard<std::mutex> lk(side->mu);
      if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess) return ENF_ELAUNCH;
      if ((rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, side->s))) return rc;
      if (hipMemsetAsync(F(W.dlt), 0, sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D), side->s) != hipSuccess) return ENF_ELAUNCH;
      if (hipEventRecord(side->join, side->s) != hipSuccess) return ENF_ELAUNCH;
      side->pending = true;
    }
  }
  const bool tsave = (stages & ENF_STAGE_TAIL_SAVE) != 0;      // stash the tail's pre-activations for the backward that follows
  if ((stages & ENF_STAGE_TAIL) &&
      (rc = enf_launch_tail(m, L, blob, yb, out, nullptr, nullptr, nullptr, tsave ? F(W.tail_act) : nullptr, 0, tsave ? 1 : 0, st)))
    return rc;
  return ENF_OK;
}

extern "C" int enf_forward(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a,
                           const float* sigma, const void* packed, float* out, float* ybar, float* lse, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return enf_forward_stages(d, x, x_bstride, p, a, sigma, packed, out, ybar, lse, workspace, workspace_bytes,
                            ENF_STAGE_PROLOGUE | ENF_STAGE_FOLD | ENF_STAGE_PAIR | ENF_STAGE_TAIL, stream);
}

extern "C" int enf_backward_latents(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a,
                                    const float* sigma, const void* packed, const float* ybar, const float* lse,
                                    const float* dout, float* dp, float* da, float* dsigma, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return enf_backward_latents_ex(d, x, x_bstride, p, a, sigma, packed, ybar, lse, dout, dp, da, dsigma, workspace,
                                 workspace_bytes, 0u, stream);
}

extern "C" int enf_backward_latents_ex(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a,
                                       const float* sigma, const void* packed, const float* ybar, const float* lse,
                                       const float* dout, float* dp, float* da, float* dsigma, void* workspace,
                                       size_t workspace_bytes, unsigned flags, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !p || !a || !packed || !ybar || !lse || !dout || !dp || !da || !dsigma || !workspace) return ENF_EINVAL;
  if (d->use_window && !sigma) return ENF_EINVAL;
  const EnfDims m = enf_dims(d);
  const EnfLayout L = enf_layout(m);
  const EnfWorkspace W = enf_workspace(m);
  if (workspace_bytes < W.total) return ENF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  const char* blob = (const char*)packed;
  // the latent table is recomputed (cheap) so the call does not depend on workspace contents, unless the
  // caller vouches that nothing has touched the workspace since the matching enf_forward
  if (!(flags & ENF_BWD_REUSE_PROLOGUE) && (rc = enf_launch_prologue(m, L, blob, p, a, sigma, F(W.lt), F(W.an), F(W.kv), st)))
    return rc;
  // z-fold backward: the per-latent matrices depend on the latent table only, so enf_wz_kernel runs on a side stream
  // (fork / join by events) beside the tail backward instead of in front of the pair kernel
  const bool zb = enf_use_zfold_bwd(m);
  SideStream* side = zb ? side_stream() : nullptr;
  bool prepared = false;
  if (zb && side && (flags & ENF_BWD_REUSE_PREPARED) && (flags & ENF_BWD_REUSE_PROLOGUE)) {
    std::lock_guard<std::mutex> lk(side->mu);
    prepared = side->pending;          // launched by the matching forward (ENF_STAGE_PREPARE_BWD)
  }
  if (!prepared && (rc = side_join_pending(st))) return rc;
  if (zb && !prepared) {
    if (side) {
      std::lock_guard<std::mutex> lk(side->mu);
      if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess) return ENF_ELAUNCH;
      if ((rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, side->s))) return rc;
      if (hipEventRecord(side->join, side->s) != hipSuccess) return ENF_ELAUNCH;
    } else if ((rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, st))) return rc;
  }
  const bool treuse = (flags & ENF_BWD_REUSE_TAIL) && (flags & ENF_BWD_REUSE_PROLOGUE);
  if ((rc = enf_launch_tail(m, L, blob, ybar, nullptr, dout, F(W.dybar), F(W.delta), F(W.tail_act), 1, treuse ? 1 : 0, st))) return rc;
  if (!prepared && hipMemsetAsync(F(W.dlt), 0, sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D), st) != hipSuccess) return ENF_ELAUNCH;
  if (zb && side) {
    std::lock_guard<std::mutex> lk(side->mu);
    if (hipStreamWaitEvent(st, side->join, 0) != hipSuccess) return ENF_ELAUNCH;
    side->pending = false;
  }
  if ((rc = enf_launch_pair_bwd(m, L, blob, x, x_bstride, F(W.lt), lse, F(W.dybar), F(W.delta), F(W.dlt), nullptr,
                                zb ? ws + W.wzt : nullptr, zb ? F(W.wzb) : nullptr, nullptr, st))) return rc;
  if ((rc = enf_launch_prologue_bwd(m, L, blob, p, sigma, F(W.an), F(W.kv), F(W.dlt), dp, da, dsigma, st))) return rc;
  return ENF_OK;
}

extern "C" int enf_lt_layout(const EnfDesc* d, int* stride, int* off_u, int* off_v0, int* off_pose, int* off_wcoef, int* off_c) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (stride) *stride = enf_lt_stride(d->H, d->D);
  if (off_u) *off_u = enf_lt_off_u(d->H, d->D);
  if (off_v0) *off_v0 = enf_lt_off_v0(d->H, d->D);
  if (off_pose) *off_pose = enf_lt_off_pose(d->H, d->D);
  if (off_wcoef) *off_wcoef = enf_lt_off_wcoef(d->H, d->D);
  if (off_c) *off_c = enf_lt_off_c(d->H, d->D);
  return ENF_OK;
}

extern "C" int enf_lt_layout_ext(const EnfDesc* d, int* off_ext, int* off_phase_q, int* off_phase_v) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (off_ext) *off_ext = enf_lt_off_ext(d->H, d->D);
  if (off_phase_q) *off_phase_q = enf_lt_off_phq(d->H, d->D);
  if (off_phase_v) *off_phase_v = enf_lt_off_phv(d->H, d->D);
  return ENF_OK;
}

static int g_zfold_mode = -2;
int enf_zfold_mode() {
  if (g_zfold_mode == -2) {
    const char* e = getenv("ENF_ZFOLD");
    g_zfold_mode = !e ? -1 : (e[0] == '0' ? 0 : 1);
  }
  return g_zfold_mode;
}
extern "C" void enf_set_zfold(int mode) { g_zfold_mode = mode < 0 ? -1 : (mode ? 1 : 0); }
static int g_zfold_bwd_mode = -2;
int enf_zfold_bwd_mode() {
  if (g_zfold_bwd_mode == -2) {
    const char* e = getenv("ENF_ZFOLD_BWD");
    g_zfold_bwd_mode = !e ? -1 : (e[0] == '0' ? 0 : 1);
  }
  return g_zfold_bwd_mode;
}
extern "C" void enf_set_zfold_bwd(int mode) { g_zfold_bwd_mode = mode < 0 ? -1 : (mode ? 1 : 0); }

extern "C" size_t enf_pair_scratch_bytes(const EnfDesc* d) {
  if (enf_check_desc(d)) return 0;
  const EnfDims m = enf_dims(d);
  if (!enf_use_zfold(m)) return 0;
  return enf_align((size_t)m.B * m.Z * m.H * enf_panel_bytes(m.D, m.D, m.bf16)) +
         enf_align(sizeof(float) * (size_t)m.B * m.Z * m.HD) + (size_t)m.B * m.Z * enf_wzu_bytes(m.H, m.D);
}

extern "C" int enf_pair_forward(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                                float* ybar, float* lse, void* scratch, size_t scratch_bytes, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !lt || !packed || !ybar || !lse) return ENF_EINVAL;
  const EnfDims m = enf_dims(d);
  const size_t need = enf_pair_scratch_bytes(d);
  if (need && (!scratch || scratch_bytes < need)) return ENF_EWORKSPACE;
  char* wz = need ? (char*)scratch : nullptr;
  float* wzb = need ? reinterpret_cast<float*>(wz + enf_align((size_t)m.B * m.Z * m.H * enf_panel_bytes(m.D, m.D, m.bf16))) : nullptr;
  char* wzu = need ? reinterpret_cast<char*>(wzb) + enf_align(sizeof(float) * (size_t)m.B * m.Z * m.HD) : nullptr;
  return enf_launch_pair_fwd(m, enf_layout(m), (const char*)packed, x, x_bstride, lt, ybar, lse, wz, wzb, wzu, 1, 1,
                             (hipStream_t)stream);
}

extern "C" void enf_pair_fwd_set_masks(unsigned* masks, int mode, int mask_B);
extern "C" void enf_pair_bwd_set_masks(const unsigned* masks, int mask_B);
extern "C" size_t enf_relu_mask_bytes(const EnfDesc* d) {
  if (enf_check_desc(d) != ENF_OK) return 0;
  return (size_t)d->B * d->Z * ((d->N + 15) / 16) * 2 * 64 * sizeof(unsigned);
}
extern "C" int enf_set_relu_masks(void* masks, int mode, int mask_signals) {
  if (mode < 0 || mode > 2 || (mode && !masks)) return ENF_EINVAL;
  enf_pair_fwd_set_masks(mode ? (unsigned*)masks : nullptr, mode, mask_signals);
  enf_pair_bwd_set_masks(mode == 2 ? (const unsigned*)masks : nullptr, mask_signals);
  return ENF_OK;
}

extern "C" int enf_pair_backward(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                                 const float* lse, const float* dybar, const float* delta, float* dlt, void* const* store,
                                 void* stream) {
  return enf_pair_backward_ex(d, x, x_bstride, lt, packed, lse, dybar, delta, dlt, store, nullptr, stream);
}

extern "C" int enf_pair_backward_ex(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                                    const float* lse, const float* dybar, const float* delta, float* dlt, void* const* store,
                                    float* dx, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !lt || !packed || !lse || !dybar || !delta || !dlt) return ENF_EINVAL;
  const EnfDims m = enf_dims(d);
  if (store)
    for (int i = 0; i < ENF_NUM_STORE(m.H); ++i)
      if (!store[i]) return ENF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dlt, 0, sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D), st) != hipSuccess) return ENF_ELAUNCH;
  return enf_launch_pair_bwd(m, enf_layout(m), (const char*)packed, x, x_bstride, lt, lse, dybar, delta, dlt, store, nullptr, nullptr, dx, st);
}
