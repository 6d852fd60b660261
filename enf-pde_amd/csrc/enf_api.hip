// enf_api.hip -- the C-ABI of include/enf_hip.h: validation, workspace carving, kernel sequencing.
// No device allocation, no host synchronisation, no settings: a call depends on its arguments only.  The host-side
// bookkeeping that exists -- one side stream per device and the pending side-stream work per (device, workspace) -- is
// keyed by what the caller passes, so calls on different workspaces / streams / devices / threads do not interact.
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include "enf_layout.h"

extern "C" {
int enf_launch_prologue(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*, float*,
                        float*, float*, hipStream_t);
int enf_launch_prologue_bwd(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, const float*,
                            const float*, const float*, float*, float*, float*, hipStream_t);
int enf_launch_pair_fwd(const EnfDims&, const EnfLayout&, const char*, const float*, long long, const float*, float*, float*,
                        char*, float*, char*, float*, int, int, hipStream_t);
int enf_launch_pair_bwd(const EnfDims&, const EnfLayout&, const char*, const float*, long long, const float*, const float*,
                        const float*, const float*, float*, void* const*, const char*, const float*, float*, hipStream_t);
int enf_launch_wz(const EnfDims&, const EnfLayout&, const char*, const float*, char*, float*, char*, char*, hipStream_t);
int enf_launch_tail(const EnfDims&, const EnfLayout&, const char*, const float*, float*, const float*, float*, float*, float*,
                    int, int, hipStream_t);
int enf_launch_tail_loss(const EnfDims&, const EnfLayout&, const char*, const float*, const float*, float, float*, float*, float*,
                         float*, hipStream_t);
}

size_t enf_xtd_part_bytes(const EnfDims& m, long long P);
int enf_launch_xtd(const EnfDims& m, void* const* store, long long P, float* const* dpair, float* part, int accumulate,
                   hipStream_t st);

extern "C" int enf_abi_version(void) { return ENF_ABI_VERSION; }

extern "C" const char* enf_strerror(int code) {
  switch (code) {
    case ENF_OK: return "ok";
    case ENF_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case ENF_EINVARIANT: return "Unknown invariant type";
    case ENF_EUNSUPPORTED: return "shape not in the compiled kernel set (num_hidden 64 or 128 after padding; num_heads 1, 2, or 4 at num_hidden 64; "
                                  "num_out <= 32; ball / ball_lat at num_hidden 64 only)";
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
  if (d->invariant_id == ENF_INV_PONITA_FULL && d->dx != 3) return ENF_EDIM;   // queries (pos_x, pos_y, theta)
  if (enf_inv_has_phase(d->invariant_id) && d->D != 64) return ENF_EUNSUPPORTED;   // (the 128-wide backward kernel has no LDS left)
  if (enf_inv_has_phase(d->invariant_id) && d->dx != 3) return ENF_EDIM;   // ball, ball_lat: (phi, theta, r) coordinates
  if (!(d->D == 64 || d->D == 128)) return ENF_EUNSUPPORTED;
  if (d->d_true < 0 || d->d_true > d->D || (d->d_true & 1)) return ENF_EINVAL;
  if (!(d->H == 1 || d->H == 2 || (d->H == 4 && d->D == 64))) return ENF_EUNSUPPORTED;   // 4 heads: 64-wide kernels only
  if (d->h_true < 0 || d->h_true > d->H) return ENF_EINVAL;
  if (d->O > 32) return ENF_EUNSUPPORTED;
  if (d->precision != ENF_PREC_F32 && d->precision != ENF_PREC_BF16) return ENF_EINVAL;
  if (d->pair_fwd_variant < ENF_VARIANT_AUTO || d->pair_fwd_variant > ENF_VARIANT_ZFOLD_ZSPLIT) return ENF_EINVAL;
  if (d->pair_bwd_variant < ENF_VARIANT_AUTO || d->pair_bwd_variant > ENF_VARIANT_ZFOLD) return ENF_EINVAL;
  if (d->mask_mode < ENF_MASK_OFF || d->mask_mode > ENF_MASK_READ || d->mask_signals < 0) return ENF_EINVAL;
  if (d->mask_mode != ENF_MASK_OFF && !d->relu_masks) return ENF_EINVAL;
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

// Side streams.  Work that can overlap the caller's stream (the z-fold backward's per-latent matrices) runs on ONE side
// stream per device, created at the first call that needs it on that device (an -DENF_AB_SWITCHES build can disable it with
// ENF_SIDE_STREAM=0).  Fork / join is by events, so the caller's stream order is preserved.  What a forward leaves pending for
// its backward (ENF_STAGE_PREPARE_BWD) is recorded against the WORKSPACE it was prepared in, with an event of its own:
// only a call on that workspace sees it.  One mutex per device keeps host threads from interleaving their record / wait
// pairs on the shared fork event.
struct SidePending { hipEvent_t join = nullptr; bool pending = false; };
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr;
  std::mutex mu;
  std::unordered_map<const void*, SidePending> ws;     // by workspace base address
  // the entry of a workspace (created on demand; entries without pending work are recycled beyond 64 workspaces)
  SidePending* entry(const void* workspace) {
    auto it = ws.find(workspace);
    if (it != ws.end()) return &it->second;
    if (ws.size() >= 64)
      for (auto j = ws.begin(); j != ws.end();) {
        if (!j->second.pending) { (void)hipEventDestroy(j->second.join); j = ws.erase(j); } else ++j;
      }
    SidePending e;
    if (hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    return &ws.emplace(workspace, e).first->second;
  }
};
static constexpr int ENF_MAX_DEVICES = 64;
static SideStream* side_stream() {      // of the calling thread's current device, or nullptr
  static std::mutex table_mu;
  static SideStream* table[ENF_MAX_DEVICES] = {};
  static bool tried[ENF_MAX_DEVICES] = {};
#ifdef ENF_AB_SWITCHES     // A/B builds only (scripts/build_variant.sh NAME -DENF_AB_SWITCHES): the product library reads no environment
  static const bool enabled = [] { const char* e = getenv("ENF_SIDE_STREAM"); return !(e && e[0] == '0'); }();
#else
  constexpr bool enabled = true;
#endif
  int dev = 0;
  if (!enabled || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ENF_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(table_mu);
  if (!tried[dev]) {
    tried[dev] = true;
    SideStream* t = new SideStream();
    if (hipStreamCreateWithFlags(&t->s, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&t->fork, hipEventDisableTiming) == hipSuccess)
      table[dev] = t;
    else
      delete t;
  }
  return table[dev];
}

// join the side-stream work an earlier ENF_STAGE_PREPARE_BWD left pending on THIS workspace (no matching backward
// came) before `st` touches the regions it writes
int enf_side_join_pending(hipStream_t st, const void* workspace);
static int side_join_pending(hipStream_t st, const void* workspace) { return enf_side_join_pending(st, workspace); }
int enf_side_join_pending(hipStream_t st, const void* workspace) {
  SideStream* side = side_stream();
  if (!side) return 0;
  std::lock_guard<std::mutex> lk(side->mu);
  auto it = side->ws.find(workspace);
  if (it != side->ws.end() && it->second.pending) {
    if (hipStreamWaitEvent(st, it->second.join, 0) != hipSuccess) return ENF_ELAUNCH;
    it->second.pending = false;
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
  // ENF_STAGE_YBAR_HALF: the hand-off to the tail as bf16 (the launchers' flag bit 1)
  const bool yhalf = (stages & ENF_STAGE_YBAR_HALF) && m.bf16 && !ybar && (stages & ENF_STAGE_PAIR) && (stages & ENF_STAGE_TAIL) &&
                     !(stages & ENF_STAGE_TAIL_SAVE) && enf_zfold_split(m) <= 1;
  if ((rc = side_join_pending(st, workspace))) return rc;
  if ((stages & ENF_STAGE_PROLOGUE) && (rc = enf_launch_prologue(m, L, blob, p, a, sigma, F(W.lt), F(W.an), F(W.kv), st))) return rc;
  if ((stages & (ENF_STAGE_PAIR | ENF_STAGE_FOLD)) &&
      (rc = enf_launch_pair_fwd(m, L, blob, x, x_bstride, F(W.lt), yb, ls, zf ? ws + W.wz : nullptr, zf ? F(W.wzb) : nullptr,
                                zf ? ws + W.wzu : nullptr, enf_zfold_split(m) > 1 ? F(W.ysplit) : nullptr, (stages & ENF_STAGE_FOLD) != 0,
                                ((stages & ENF_STAGE_PAIR) != 0 ? 1 : 0) | (yhalf ? 2 : 0), st)))
    return rc;
  if ((stages & ENF_STAGE_PREPARE_BWD) && enf_use_zfold_bwd(m)) {
    // what the backward needs from the latent table alone -- its per-latent folded matrices, the zeroed gradient table --
    // starts on the side stream behind the pair kernel, beside the tail, the caller's loss and the tail backward
    SideStream* side = side_stream();
    if (side) {
      std::lock_guard<std::mutex> lk(side->mu);
      SidePending* e = side->entry(workspace);
      if (e) {
        if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess) return ENF_ELAUNCH;
        if ((rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, side->s))) return rc;
        if (hipMemsetAsync(F(W.dlt), 0, sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D), side->s) != hipSuccess) return ENF_ELAUNCH;
        if (hipEventRecord(e->join, side->s) != hipSuccess) return ENF_ELAUNCH;
        e->pending = true;
      }
    }
  }
  const bool tsave = (stages & ENF_STAGE_TAIL_SAVE) != 0;      // stash the tail's pre-activations for the backward that follows
  if ((stages & ENF_STAGE_TAIL) &&
      (rc = enf_launch_tail(m, L, blob, yb, out, nullptr, nullptr, nullptr, tsave ? F(W.tail_act) : nullptr, 0, (tsave ? 1 : 0) | (yhalf ? 2 : 0), st)))
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
  const bool zb = enf_use_zfold_bwd(m);
  if (flags & ENF_BWD_ONLY_PAIR) {      // measurement hook: the pair kernel alone, on what a complete backward left behind
    if ((rc = side_join_pending(st, workspace))) return rc;
    if (hipMemsetAsync(F(W.dlt), 0, sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D), st) != hipSuccess) return ENF_ELAUNCH;
    return enf_launch_pair_bwd(m, L, blob, x, x_bstride, F(W.lt), lse, F(W.dybar), F(W.delta), F(W.dlt), nullptr,
                               zb ? ws + W.wzt : nullptr, zb ? F(W.wzb) : nullptr, nullptr, st);
  }
  // the latent table is recomputed (cheap) so the call does not depend on workspace contents, unless the
  // caller vouches that nothing has touched the workspace since the matching enf_forward
  if (!(flags & ENF_BWD_REUSE_PROLOGUE) && (rc = enf_launch_prologue(m, L, blob, p, a, sigma, F(W.lt), F(W.an), F(W.kv), st)))
    return rc;
  // z-fold backward: the per-latent matrices depend on the latent table only, so enf_wz_kernel runs on a side stream
  // (fork / join by events) beside the tail backward instead of in front of the pair kernel
  SideStream* side = zb ? side_stream() : nullptr;
  bool prepared = false;
  if (zb && side && (flags & ENF_BWD_REUSE_PREPARED) && (flags & ENF_BWD_REUSE_PROLOGUE)) {
    std::lock_guard<std::mutex> lk(side->mu);
    auto it = side->ws.find(workspace);
    prepared = it != side->ws.end() && it->second.pending;   // launched by the matching forward ON THIS WORKSPACE
  }
  if (!prepared && (rc = side_join_pending(st, workspace))) return rc;
  if (zb && !prepared) {
    bool forked = false;
    if (side) {
      std::lock_guard<std::mutex> lk(side->mu);
      SidePending* e = side->entry(workspace);
      if (e) {
        if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess) return ENF_ELAUNCH;
        if ((rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, side->s))) return rc;
        if (hipEventRecord(e->join, side->s) != hipSuccess) return ENF_ELAUNCH;
        e->pending = true;              // until joined below: a failure in between leaves it for the next call's join
        forked = true;
      }
    }
    if (!forked && (rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, st))) return rc;
  }
  const bool treuse = (flags & ENF_BWD_REUSE_TAIL) && (flags & ENF_BWD_REUSE_PROLOGUE);
  if ((rc = enf_launch_tail(m, L, blob, ybar, nullptr, dout, F(W.dybar), F(W.delta), F(W.tail_act), 1, treuse ? 1 : 0, st))) return rc;
  if (!prepared && hipMemsetAsync(F(W.dlt), 0, sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D), st) != hipSuccess) return ENF_ELAUNCH;
  if ((rc = side_join_pending(st, workspace))) return rc;      // the per-latent matrices (and, if prepared, the zeroed table)
  if ((rc = enf_launch_pair_bwd(m, L, blob, x, x_bstride, F(W.lt), lse, F(W.dybar), F(W.delta), F(W.dlt), nullptr,
                                zb ? ws + W.wzt : nullptr, zb ? F(W.wzb) : nullptr, nullptr, st))) return rc;
  if ((rc = enf_launch_prologue_bwd(m, L, blob, p, sigma, F(W.an), F(W.kv), F(W.dlt), dp, da, dsigma, st))) return rc;
  return ENF_OK;
}


// One inner step of the MAML loop in one call (SURVEY.md 8d's unit of work; pde_trainer.py:175-207): forward on the sampled points,
// mean squared error against `target` and its gradient, backward to the latents.  Same kernels as enf_forward_stages +
// enf_mse_value_grad + enf_backward_latents_ex with every REUSE flag, except that the tail runs ONCE (forward chain, loss and
// backward chain in one kernel, enf_tail.hip: LOSS) and neither `out` nor `d out` exists.
extern "C" int enf_fit_step(const EnfDesc* d, const float* x, int64_t x_bstride, const float* p, const float* a, const float* sigma,
                            const void* packed, const float* target, float grad_scale, float* loss, float* dp, float* da,
                            float* dsigma, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !p || !a || !packed || !target || !loss || !dp || !da || !dsigma || !workspace) return ENF_EINVAL;
  if (d->use_window && !sigma) return ENF_EINVAL;
  const EnfDims m = enf_dims(d);
  const EnfLayout L = enf_layout(m);
  const EnfWorkspace W = enf_workspace(m);
  if (workspace_bytes < W.total) return ENF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  const char* blob = (const char*)packed;
  const bool zf = enf_use_zfold(m), zb = enf_use_zfold_bwd(m);
  if ((rc = side_join_pending(st, workspace))) return rc;
  if ((rc = enf_launch_prologue(m, L, blob, p, a, sigma, F(W.lt), F(W.an), F(W.kv), st))) return rc;
  if ((rc = enf_launch_pair_fwd(m, L, blob, x, x_bstride, F(W.lt), F(W.ybar), F(W.lse), zf ? ws + W.wz : nullptr, zf ? F(W.wzb) : nullptr,
                                zf ? ws + W.wzu : nullptr, enf_zfold_split(m) > 1 ? F(W.ysplit) : nullptr, 1, 1, st)))
    return rc;
  // what the backward pair kernel needs from the latent table alone runs on the side stream beside the tail
  const size_t dlt_bytes = sizeof(float) * (size_t)m.B * m.Z * enf_lt_stride(m.H, m.D);
  bool forked = false;
  SideStream* side = zb ? side_stream() : nullptr;
  if (side) {
    std::lock_guard<std::mutex> lk(side->mu);
    SidePending* e = side->entry(workspace);
    if (e) {
      if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess) return ENF_ELAUNCH;
      if ((rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, side->s))) return rc;
      if (hipMemsetAsync(F(W.dlt), 0, dlt_bytes, side->s) != hipSuccess) return ENF_ELAUNCH;
      if (hipEventRecord(e->join, side->s) != hipSuccess) return ENF_ELAUNCH;
      e->pending = true;
      forked = true;
    }
  }
  if (!forked) {
    if (zb && (rc = enf_launch_wz(m, L, blob, F(W.lt), nullptr, F(W.wzb), nullptr, ws + W.wzt, st))) return rc;
    if (hipMemsetAsync(F(W.dlt), 0, dlt_bytes, st) != hipSuccess) return ENF_ELAUNCH;
  }
  if ((rc = enf_launch_tail_loss(m, L, blob, F(W.ybar), target, grad_scale, loss, F(W.dybar), F(W.delta), F(W.tail_act), st))) return rc;
  if ((rc = side_join_pending(st, workspace))) return rc;
  if ((rc = enf_launch_pair_bwd(m, L, blob, x, x_bstride, F(W.lt), F(W.lse), F(W.dybar), F(W.delta), F(W.dlt), nullptr,
                                zb ? ws + W.wzt : nullptr, zb ? F(W.wzb) : nullptr, nullptr, st))) return rc;
  return enf_launch_prologue_bwd(m, L, blob, p, sigma, F(W.an), F(W.kv), F(W.dlt), dp, da, dsigma, st);
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

// What ENF_VARIANT_AUTO resolves to is a function of the shape alone (enf_layout.h); per call the caller chooses with
// EnfDesc.pair_fwd_variant / pair_bwd_variant.  Only an -DENF_AB_SWITCHES build (A/B runs of a whole program) looks at
// ENF_ZFOLD / ENF_ZFOLD_BWD in the environment (read once).
int enf_zfold_env(int backward) {
#ifdef ENF_AB_SWITCHES
  static const int mode[2] = {
      [] { const char* e = getenv("ENF_ZFOLD"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }(),
      [] { const char* e = getenv("ENF_ZFOLD_BWD"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }()};
  return mode[backward ? 1 : 0];
#else
  (void)backward;
  return -1;
#endif
}

extern "C" int enf_pair_variant(const EnfDesc* d, int backward) {
  const int rc = enf_check_desc(d);
  if (rc) return rc;
  const EnfDims m = enf_dims(d);
  if (backward) return enf_use_zfold_bwd(m) ? ENF_VARIANT_ZFOLD : ENF_VARIANT_LATENT_SPLIT;
  const int sp = enf_zfold_split(m);
  return sp > 1 ? ENF_VARIANT_ZFOLD_ZSPLIT : (sp == 1 ? ENF_VARIANT_ZFOLD : ENF_VARIANT_LATENT_SPLIT);
}

extern "C" int enf_pair_partition(const EnfDesc* d, int32_t* run, int32_t* workgroups, int32_t* parts) {
  const int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!run || !workgroups || !parts) return ENF_EINVAL;
  const EnfDims m = enf_dims(d);
  if (enf_zfold_split(m) < 2) return 0;
  const EnfStreamK k = enf_zfold_streamk(m);
  *run = k.len; *workgroups = k.wgs; *parts = k.parts;
  return 1;
}

extern "C" size_t enf_pair_scratch_bytes(const EnfDesc* d) {
  if (enf_check_desc(d)) return 0;
  const EnfDims m = enf_dims(d);
  if (!enf_use_zfold(m)) return 0;
  return enf_align((size_t)m.B * m.Z * m.H * enf_panel_bytes(m.D, m.D, m.bf16)) +
         enf_align(sizeof(float) * (size_t)m.B * m.Z * m.HD) + enf_align((size_t)m.B * m.Z * enf_wzu_bytes(m.H, m.D)) +
         (enf_zfold_split(m) > 1 ? sizeof(float) * enf_zfold_split(m) * ((size_t)m.B * m.N * m.HD + (size_t)m.B * m.N * m.H * 3) : 0);
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
  float* ysplit = need && enf_zfold_split(m) > 1 ? reinterpret_cast<float*>(wzu + enf_align((size_t)m.B * m.Z * enf_wzu_bytes(m.H, m.D))) : nullptr;
  return enf_launch_pair_fwd(m, enf_layout(m), (const char*)packed, x, x_bstride, lt, ybar, lse, wz, wzb, wzu, ysplit, 1, 1,
                             (hipStream_t)stream);
}

extern "C" size_t enf_relu_mask_bytes(const EnfDesc* d) {
  if (enf_check_desc(d) != ENF_OK) return 0;
  const size_t signals = d->mask_signals > 0 ? d->mask_signals : d->B;
  return signals * d->Z * ((d->N + 15) / 16) * 2 * 64 * sizeof(unsigned);
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

// ---- weight gradients of the per-pair chain: K3 (STORE) -> K4 (enf_xtd.hip), chunked over signals
static size_t bw_store_bytes(const EnfDims& m, int cb) {
  return enf_align((size_t)cb * m.Z * m.N * m.D * (m.bf16 ? 2 : 4));          // one ENF_S_* buffer of a chunk
}
static size_t bw_scratch_bytes(const EnfDims& m, int cb) {
  return (size_t)ENF_NUM_STORE(m.H) * bw_store_bytes(m, cb) + enf_xtd_part_bytes(m, (long long)cb * m.Z * m.N);
}
extern "C" size_t enf_backward_weights_scratch_bytes(const EnfDesc* d, int chunk_signals) {
  if (enf_check_desc(d) != ENF_OK || chunk_signals < 1 || chunk_signals > d->B) return 0;
  return bw_scratch_bytes(enf_dims(d), chunk_signals);
}

extern "C" int enf_backward_weights(const EnfDesc* d, const float* x, int64_t x_bstride, const float* lt, const void* packed,
                                    const float* lse, const float* dybar, const float* delta, float* dlt,
                                    float* const* dpair, float* dx, void* scratch, size_t scratch_bytes, void* stream) {
  int rc = enf_check_desc(d);
  if (rc) return rc;
  if (!x || !lt || !packed || !lse || !dybar || !delta || !dlt || !dpair || !scratch) return ENF_EINVAL;
  for (int i = 0; i < ENF_NUM_PAIR_TENSORS; ++i)
    if (i != ENF_P_COEFQ && i != ENF_P_COEFV && !dpair[i]) return ENF_EINVAL;
  EnfDims m = enf_dims(d);
  const EnfLayout L = enf_layout(m);
  // the largest chunk of signals whose store fits; with masks, whole groups of mask_signals (signal b replays b % mask_signals)
  const int step = m.mask_mode == ENF_MASK_READ && m.mask_B < m.B ? m.mask_B : 1;
  int cb = m.B;
  while (cb > step && bw_scratch_bytes(m, cb) > scratch_bytes) cb = (cb - 1) / step * step;
  if (cb < 1 || bw_scratch_bytes(m, cb) > scratch_bytes) return ENF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int stride = enf_lt_stride(m.H, m.D);
  if (hipMemsetAsync(dlt, 0, sizeof(float) * (size_t)m.B * m.Z * stride, st) != hipSuccess) return ENF_ELAUNCH;
  char* sc = (char*)scratch;
  void* store[ENF_NUM_STORE(4)];
  const size_t sb = bw_store_bytes(m, cb);
  for (int i = 0; i < ENF_NUM_STORE(m.H); ++i) store[i] = sc + (size_t)i * sb;
  float* part = reinterpret_cast<float*>(sc + (size_t)ENF_NUM_STORE(m.H) * sb);
  const int B = m.B;
  for (int b0 = 0; b0 < B; b0 += cb) {
    const int nb = b0 + cb <= B ? cb : B - b0;
    EnfDims mc = m;
    mc.B = nb; mc.mask_b0 = b0;
    const size_t qo = (size_t)b0 * m.N;
    if ((rc = enf_launch_pair_bwd(mc, L, (const char*)packed, x + (size_t)b0 * x_bstride, x_bstride, lt + (size_t)b0 * m.Z * stride,
                                  lse + qo * m.H, dybar + qo * m.HD, delta + qo * m.H, dlt + (size_t)b0 * m.Z * stride, store,
                                  nullptr, nullptr, dx ? dx + qo * m.dx : nullptr, st)))
      return rc;
    if ((rc = enf_launch_xtd(mc, store, (long long)nb * m.Z * m.N, dpair, part, b0 > 0, st))) return rc;
  }
  return ENF_OK;
}
