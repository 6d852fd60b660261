"""Outer (meta) steps of the ENF trainer, mirroring experiments/fitting/trainers/pde_trainer.py: the nef phase
(nef_train_step), the latent-ODE phase (ode_loss, ode_train_step, dual_train_step) and val_step.

Reference:  recon_loss, grads = jax.value_and_grad(self.enf_loss)(params, state, trajectory)   (pde_trainer.py:255)
where enf_loss = the loss of the LAST inner step after S meta-SGD steps from the shared latent initialisation
(pde_trainer.py:122-235, 483-531), differentiated w.r.t. params = {nef, autodecoder, meta_sgd_lrs} THROUGH the
inner steps, followed by clip_by_global_norm(1) + AdamW on nef, Adam on the latent initialisation when
learning_rate_codes != 0, Adam on the inner learning rates and clip(lrs, 1e-6, 10)   (pde_trainer.py:258-273).

Here the decoder is once-differentiable (hand-written HIP backward), so the meta-gradient is assembled by the
adjoint recursion over the inner steps

    phi_{s+1} = phi_s - alpha * g_s(phi_s, theta),          g_s = B grad_phi L_s  (sigma masked)
    lambda_S  = grad_phi L_S(phi_S),   g_theta = grad_theta L_S(phi_S)
    lambda_s  = lambda_{s+1} - B H_s w,      g_theta -= B d/dtheta[grad_phi L_s . w],      w = alpha * lambda_{s+1}
    g_alpha  -= sum lambda_{s+1} * g_s

whose only second-order objects are Hessian(-mixed)-vector products along ONE direction w per step.
``second_order="fd"`` evaluates them by central differences of first-order gradients at phi_s +- eps w (one extra
training-path pass per inner step over a batch of 2B signals: the +eps and -eps latent sets side by side, the second
half entering the loss negated, so that the pass returns the difference of the two gradients; ``fd_step`` = the largest component of the
perturbation); ``second_order="none"`` drops them (first-order MAML).

Accuracy against exact double-backward of the oracle (scripts/meta_grad_err.py, tests/test_gpu_trainer.py; f32
mode): first-order MAML is off by 7-50 % per tensor on the test problems -- the second-order terms matter.  Plain
finite differences of the gradients brought 42 of the 46 weight tensors to ~1e-3..1e-2 but left the four tensors feeding
a relu (layers_0 kernel / bias of the two RFFNets) and the position initialisation at 10-30 % FOR EVERY STEP SIZE: a
difference of first-order gradients converges to the DISTRIBUTIONAL second derivative, in which every relu unit whose
sign changes between phi_s - eps w and phi_s + eps w contributes a finite amount (their number scales with eps, each
one's weight with 1/eps), while automatic differentiation -- the reference's jax.grad, the oracle -- has relu'' = 0.
(scripts/meta_grad_fd_oracle.py reproduces this with the CPU oracle in fp64: same 22 % with free masks, 1e-3 -> 1e-8
as the step shrinks with frozen masks.)  So the two perturbed passes run with the relu masks FROZEN at phi_s
(``freeze_relu``; include/enf_hip.h: enf_set_relu_masks -- the forward sweep's own pair-kernel forward at phi_s records
them, the forward and the weight-gradient backward of the perturbed pass replay them): every tensor, the latent initialisation
and the inner learning rates then agree with exact second-order autograd to 1e-3 at fd_step 2e-2, 1e-4 at 5e-3 (the
f32 default) and 3-7e-5 at 1e-3.  bf16 kernels (scripts/meta_grad_err_bf16.py): best at 2e-2 (their default): median 7e-3
per tensor, 4-8 % on the worst (the bf16 noise floor of the first-order weight gradients themselves).
"""
import math
from dataclasses import dataclass, field
from types import SimpleNamespace

import torch

from ..inner_loop import _pose, make_masks, inner_loop, decode
from .trainer_utils.solvers import solve_latent_ode
from ..optim import Adam, AdamW, clip_by_global_norm
from ..parallel import allreduce_mean_
from ...enf.models import TENSOR_PATHS, BLOCK_PATHS, tensor_paths, _get, _set

LATENT_KEYS = ("p_pos", "p_ori", "a", "gaussian_window")


def _leaves(tree):
    """Leaves of a nested parameter dict in a fixed (sorted-key) order."""
    out = []
    for k in sorted(tree):
        out += _leaves(tree[k]) if isinstance(tree[k], dict) else [tree[k]]
    return out


def _unflatten(tree, leaves):
    it = iter(leaves)

    def build(t):
        return {k: (build(t[k]) if isinstance(t[k], dict) else next(it)) for k in sorted(t)}
    return build(tree)


def _tree_from_tensors(tensors):
    out = {}
    layers = (len(tensors) - len(TENSOR_PATHS)) // len(BLOCK_PATHS)       # self-attention blocks, if any
    for path, t in zip(tensor_paths(layers), tensors):
        _set(out, path, t)
    return {"params": out}


def _loss(nef, params, coords, img, masks, s, lat):
    B = img.shape[0]
    n_ori = nef.cross_attn_invariant.num_z_ori_dims
    xs = coords[masks[:, s]][None].expand(B, -1, -1)                    # pde_trainer.py:193-197
    ys = img[:, masks[:, s]]
    out = nef.apply(params, xs, _pose(lat, n_ori), lat["a"], lat.get("gaussian_window"))
    return ((out - ys) ** 2).mean()                                     # pde_trainer.py:185


def _latent_grads(nef, params, coords, img, masks, s, lat, keys):
    leaves = {k: lat[k].detach().requires_grad_(True) for k in lat}
    g = torch.autograd.grad(_loss(nef, params, coords, img, masks, s, leaves), [leaves[k] for k in keys], allow_unused=True)
    return {k: (torch.zeros_like(lat[k]) if gk is None else gk) for k, gk in zip(keys, g)}


def _full_grads(nef, weights, coords, img, masks, s, lat, keys):
    """(loss, grads w.r.t. the 46 weight tensors, grads w.r.t. the latents) on the training path."""
    w = [t.detach().requires_grad_(True) for t in weights]
    leaves = {k: lat[k].detach().requires_grad_(True) for k in lat}
    loss = _loss(nef, _tree_from_tensors(w), coords, img, masks, s, leaves)
    g = torch.autograd.grad(loss, w + [leaves[k] for k in keys], allow_unused=True)
    gw = [torch.zeros_like(t) if gi is None else gi for t, gi in zip(w, g[:len(w)])]
    gl = {k: (torch.zeros_like(lat[k]) if gi is None else gi) for k, gi in zip(keys, g[len(w):])}
    return loss.detach(), gw, gl


def _diff_grads(nef, weights, coords, img, masks, s, plus, minus, keys, relu_buf=None):
    """grads(plus) - grads(minus) of the step-s loss, w.r.t. the weights and the latents, in ONE training-path pass: the two
    latent sets run as one batch of 2B signals whose second half enters the loss with a minus sign (the outer step is
    bound by its many small kernels, so one pass of twice the batch costs about half of two passes)."""
    B = img.shape[0]
    w = [t.detach().requires_grad_(True) for t in weights]
    leaves = {k: torch.cat([plus[k], minus[k]], 0).detach().requires_grad_(True) for k in plus}
    n_ori = nef.cross_attn_invariant.num_z_ori_dims
    xs = coords[masks[:, s]][None].expand(2 * B, -1, -1)
    ys = img[:, masks[:, s]]
    import contextlib
    with (nef.relu_masks(relu_buf, "read", B) if relu_buf is not None else contextlib.nullcontext()):
        out = nef.apply(_tree_from_tensors(w), xs, _pose(leaves, n_ori), leaves["a"], leaves.get("gaussian_window"))
        loss = ((out[:B] - ys) ** 2).mean() - ((out[B:] - ys) ** 2).mean()
        g = torch.autograd.grad(loss, w + [leaves[k] for k in keys], allow_unused=True)
    gw = [torch.zeros_like(t) if gi is None else gi for t, gi in zip(w, g[:len(w)])]
    gl = {k: (torch.zeros_like(plus[k]) if gi is None else gi[:B] + gi[B:]) for k, gi in zip(keys, g[len(w):])}
    return gw, gl


def meta_gradients(nef, nef_params, latents0, lrs, coords, img, masks, optimize_gaussian_window=False,
                   second_order="fd", fd_step=None, noise_pos=0.0, generator=None, terminal=None, freeze_relu=True):
    """Value and gradient of the last-inner-step loss w.r.t. (nef weights, meta-init latents, inner lrs).

    Returns (loss, grads) with grads = {'nef': [46 tensors in ENF_W_* order], 'autodecoder': {key: (1,Z,.)},
    'meta_sgd_lrs': {key: like lrs[key]}}.

    ``freeze_relu``: take the finite differences with the relu masks frozen at the unperturbed latents (module docstring).
    ``terminal(weights, lat, keys) -> (loss, d loss/d weights, {key: d loss/d lat[key]})`` replaces the objective on
    the fitted latents (default: the reconstruction loss on the last mask); dual_train_step passes the roll-out loss.
    """
    if second_order not in ("fd", "none"):
        raise ValueError("second_order must be 'fd' or 'none'")
    if fd_step is None:      # truncation (~step^2) against the rounding of the first-order gradients (~1 / step): bf16 kernels
        fd_step = 2e-2 if getattr(nef, "precision", "f32") in ("bf16", "bfloat16") else 5e-3      # are 100x noisier
    B = img.shape[0]
    S = masks.shape[1] - 1
    weights = nef.param_tensors(nef_params)
    frozen = _tree_from_tensors([t.detach() for t in weights])           # inference path for the inner steps
    lat = {k: v.detach().repeat_interleave(B, dim=0).clone() for k, v in latents0.items()}
    if noise_pos:
        lat["p_pos"] = lat["p_pos"] + torch.randn(lat["p_pos"].shape, generator=generator, device="cpu").to(lat["p_pos"].device) * noise_pos
    keys = [k for k in lat if not (k == "gaussian_window" and not nef.use_gaussian_window)]

    def masked(k):          # sigma takes part in the inner update only when asked to (pde_trainer.py:210-212)
        return k == "gaussian_window" and not optimize_gaussian_window

    # ---- forward sweep, keeping every phi_s and g_s
    phis, gs, relu_bufs = [], [], []
    freeze = second_order == "fd" and freeze_relu and hasattr(nef, "relu_masks")
    import contextlib
    for s in range(S):
        # (the forward of this pass also records the relu masks at phi_s for the adjoint sweep's frozen-mask differences)
        relu_bufs.append(nef.relu_mask_buffer(B, masks.shape[0], lat["a"].shape[1], coords.device) if freeze else None)
        with (nef.relu_masks(relu_bufs[s], "write", B) if freeze else contextlib.nullcontext()):
            g = _latent_grads(nef, frozen, coords, img, masks, s, lat, keys)
        g = {k: (torch.zeros_like(lat[k]) if (k not in g or masked(k)) else g[k] * B) for k in lat}    # pde_trainer.py:207
        phis.append(lat)
        gs.append(g)
        lat = {k: (lat[k] - lrs[k] * g[k]).detach() for k in lat}                                       # pde_trainer.py:215-219
    # ---- last step: value, d/d theta, lambda_S
    if terminal is None:
        loss, g_theta, lam = _full_grads(nef, weights, coords, img, masks, S, lat, keys)
    else:
        loss, g_theta, lam = terminal(weights, lat, keys)
    lam = {k: lam.get(k, torch.zeros_like(lat[k])) for k in lat}
    g_alpha = {k: torch.zeros_like(lrs[k]) for k in lrs}
    # ---- adjoint sweep
    for s in reversed(range(S)):
        for k in lrs:
            if k in lam and not masked(k):
                prod = lam[k] * gs[s][k]
                g_alpha[k] -= prod.sum(dim=(0, 1)) if lrs[k].numel() > 1 else prod.sum().reshape(lrs[k].shape)
        if second_order == "none":
            continue
        w = {k: (torch.zeros_like(lam[k]) if masked(k) else lrs[k] * lam[k]) for k in lam}
        # the step size stays on the device (no host synchronisation inside the outer step): eps = fd_step / max|w|; a
        # direction that vanishes identically gives plus == minus, a zero difference, and a finite c below -- no contribution
        wmax = torch.stack([v.abs().max() for v in w.values()]).max().clamp_min(1e-30)
        eps = fd_step / wmax
        plus = {k: phis[s][k] + eps * w[k] for k in lam}
        minus = {k: phis[s][k] - eps * w[k] for k in lam}
        # with the relu masks AT phi_s both perturbed passes differentiate the same piecewise-linear branch, so their
        # difference is the almost-everywhere second derivative (what jax.grad of the inner steps computes) instead of
        # also counting the units that flip between phi_s - eps w and phi_s + eps w
        gw_d, gl_d = _diff_grads(nef, weights, coords, img, masks, s, plus, minus, keys, relu_bufs[s])
        c = B / (2.0 * eps)                                                     # a 0-dim device tensor
        g_theta = list(torch._foreach_sub(g_theta, torch._foreach_mul(gw_d, c)))
        lam = {k: lam[k] - c * gl_d[k] if k in gl_d else lam[k] for k in lam}
    g_lat0 = {k: lam[k].sum(dim=0, keepdim=True) for k in lam}
    return loss, {"nef": g_theta, "autodecoder": g_lat0, "meta_sgd_lrs": g_alpha}


@dataclass
class TrainState:
    params: dict
    nef_opt_state: dict
    autodecoder_opt_state: dict
    meta_sgd_opt_state: dict
    ode_opt_state: dict = None
    step: int = 0
    rng: torch.Generator = field(default_factory=lambda: torch.Generator().manual_seed(0))


class MetaSGDPDETrainer:
    """nef phase of MetaSGDPDETrainer (pde_trainer.py): init_train_state / nef_train_step.

    ``config`` carries the reference's field names: optimizer.learning_rate_enf, optimizer.learning_rate_codes,
    meta.learning_rate_meta_sgd, meta.num_inner_steps, meta.inner_learning_rate_{p,a,window},
    meta.noise_pos_inner_loop, nef.optimize_gaussian_window, training.max_num_sampled_points; with an ``ode_model``
    also optimizer.learning_rate_ode, node.dt, node.method, dataset.traj_len_train, dataset.traj_len_out_horizon.
    """

    def __init__(self, config, nef, outer_autodecoder, coords, seed=0, second_order="fd", fd_step=None, ode_model=None):
        self.config, self.nef, self.outer_autodecoder, self.coords, self.seed = config, nef, outer_autodecoder, coords, seed
        self.second_order, self.fd_step = second_order, fd_step
        self.ode_model = ode_model
        # opt-in: ode / dual train steps replay captured hipGraphs of the derivative evaluations (PonitaODEGen.graphed_train).
        # Off by default: an isolated evaluation is host-bound and gains (2.4 -> 1.8 ms), but inside a train step the host runs
        # ahead of the GPU, the eager queue issues kernels back to back, and graph replay's larger kernel-to-kernel gaps make
        # the 10-frame step SLOWER (30.0 vs 26.4 ms, scripts/bench_ode.py)
        self.graph_ode_training = bool(getattr(getattr(config, "training", None), "graph_ode_training", False))
        o, m = config.optimizer, config.meta
        self.nef_opt = AdamW(o.learning_rate_enf)                              # after clip_by_global_norm(1.0)
        self.autodecoder_opt = Adam(o.learning_rate_codes)
        self.meta_sgd_opt = Adam(m.learning_rate_meta_sgd)
        self.ode_opt = Adam(getattr(o, "learning_rate_ode", 1e-3)) if ode_model is not None else None   # pde_trainer.py:66

    def init_train_state(self, nef_params=None, ode_params=None):
        cfg, dev = self.config, self.coords.device
        g = torch.Generator().manual_seed(self.seed)
        ad = self.outer_autodecoder.init(g, device=dev)                          # pde_trainer.py:79-81
        C = ad["params"]["a"].shape[-1]
        lrs = {"p_pos": torch.ones(1, device=dev) * cfg.meta.inner_learning_rate_p,           # pde_trainer.py:83-97
               "a": torch.ones(C, device=dev) * cfg.meta.inner_learning_rate_a,
               "gaussian_window": torch.ones(1, device=dev) * cfg.meta.inner_learning_rate_window}
        if self.outer_autodecoder.num_ori_dims > 0:
            lrs["p_ori"] = torch.ones(1, device=dev) * cfg.meta.inner_learning_rate_p
        if nef_params is None:
            nef_params = self.nef.init(g, device=dev)                            # pde_trainer.py:99-102
        params = {"nef": nef_params, "autodecoder": ad, "meta_sgd_lrs": lrs}
        ode_opt_state = None
        if self.ode_model is not None:                                           # pde_trainer.py:104-105
            P = ad["params"]
            p0 = torch.cat((P["p_pos"], P["p_ori"]), -1) if self.outer_autodecoder.num_ori_dims > 0 else P["p_pos"]
            params["ode_params"] = ode_params if ode_params is not None else \
                self.ode_model.init(self.seed + 1, (p0, P["a"], P.get("gaussian_window")), device=dev)
            ode_opt_state = self.ode_opt.init(_leaves(params["ode_params"]))
        return TrainState(params=params,
                          nef_opt_state=self.nef_opt.init(self.nef.param_tensors(nef_params)),
                          autodecoder_opt_state=self.autodecoder_opt.init(list(ad["params"].values())),
                          meta_sgd_opt_state=self.meta_sgd_opt.init(list(lrs.values())),
                          ode_opt_state=ode_opt_state, step=0, rng=g)

    def save_checkpoint(self, state, path, epoch=0):
        """_base_pde_trainer.py:192-202: the whole train state (parameters, every optimiser's count / mu / nu, step, rng)
        and the config, in one .npz (enf_pde_amd/checkpoint.py: save_train_state)."""
        from ...checkpoint import save_train_state
        save_train_state(path, state, config=self.config, epoch=epoch)

    def load_checkpoint(self, path, **init_kwargs):
        """_base_pde_trainer.py:204-237: restore into a freshly initialised state of this trainer.  Returns (state, epoch)."""
        from ...checkpoint import load_train_state
        state, epoch, _ = load_train_state(path, self.init_train_state(**init_kwargs))
        return state, epoch

    def _latents0(self, state):
        P = state.params["autodecoder"]["params"]
        keys = [k for k in LATENT_KEYS if k in P and not (k == "p_ori" and self.outer_autodecoder.num_ori_dims == 0)]
        return {k: P[k] for k in keys}

    def nef_train_step(self, state, batch, masks=None):
        """One outer step on ``batch`` = (B, N, O) initial states (trajectory[:, 0], pde_trainer.py:485-487).
        Returns (recon_loss, new_state).  In a multi-rank run every rank passes its shard of the meta-batch;
        the outer gradients are averaged with one flat all-reduce before the (identical) optimiser updates."""
        cfg = self.config
        img = batch.reshape(batch.shape[0], -1, batch.shape[-1])
        if masks is None:
            masks = make_masks(self.coords.shape[0], cfg.training.max_num_sampled_points, cfg.meta.num_inner_steps,
                               generator=state.rng, device=self.coords.device)
        lat0 = self._latents0(state)
        lrs = state.params["meta_sgd_lrs"]
        loss, grads = meta_gradients(self.nef, state.params["nef"], lat0, lrs, self.coords, img, masks,
                                     optimize_gaussian_window=getattr(cfg.nef, "optimize_gaussian_window", False),
                                     second_order=self.second_order, fd_step=self.fd_step,
                                     noise_pos=getattr(cfg.meta, "noise_pos_inner_loop", 0.0), generator=state.rng)
        lat_keys, lr_keys = list(lat0.keys()), list(lrs.keys())
        flat = grads["nef"] + [grads["autodecoder"][k] for k in lat_keys] + [grads["meta_sgd_lrs"][k] for k in lr_keys] + [loss.reshape(1)]
        allreduce_mean_(flat, weight=img.shape[0])                               # one exchange per outer step
        loss = flat[-1][0]
        # nef: clip_by_global_norm(1.0) -> adamw                                  (pde_trainer.py:60-63,258-259)
        weights = self.nef.param_tensors(state.params["nef"])
        new_w, nef_opt_state = self.nef_opt.update(clip_by_global_norm(grads["nef"], 1.0), state.nef_opt_state, weights)
        nef_params = _tree_from_tensors(new_w)
        # latent initialisation: adam, only when learning_rate_codes != 0         (pde_trainer.py:261-268)
        ad = state.params["autodecoder"]
        ad_state = state.autodecoder_opt_state
        if cfg.optimizer.learning_rate_codes != 0:
            P = ad["params"]
            names = list(P.keys())
            g = [grads["autodecoder"].get(k, torch.zeros_like(P[k])) for k in names]
            new_p, ad_state = self.autodecoder_opt.update(g, ad_state, [P[k] for k in names])
            ad = {"params": dict(zip(names, new_p))}
        # inner learning rates: adam, then clip to [1e-6, 10]                      (pde_trainer.py:270-273)
        new_lrs, lr_state = self.meta_sgd_opt.update([grads["meta_sgd_lrs"][k] for k in lr_keys], state.meta_sgd_opt_state,
                                                     [lrs[k] for k in lr_keys])
        lrs = {k: v.clamp(1e-6, 10.0) for k, v in zip(lr_keys, new_lrs)}
        params = dict(state.params, nef=nef_params, autodecoder=ad, meta_sgd_lrs=lrs)     # ode_params carried over
        new_state = TrainState(params=params, nef_opt_state=nef_opt_state, autodecoder_opt_state=ad_state,
                               meta_sgd_opt_state=lr_state, ode_opt_state=state.ode_opt_state, step=state.step + 1, rng=state.rng)
        return loss, new_state

    def meta_gradient_report(self, state, batch, masks=None):
        """How far this trainer's meta-gradient (the model's own arithmetic, normally bf16) is from the same meta-gradient
        taken with f32-mode kernels on the same batch and masks: {tensor path: relative L2 difference} plus "median" / "max"
        over the weight tensors and entries for the latent initialisation ("lat0/...") and inner rates ("lr/...").  A run
        in bf16 mode can log this every so often instead of trusting the contract of tests/test_gpu_bf16_contract.py blindly
        (typical: median 7e-3, worst tensor 4-8 %, the relu layers of the two RFFNets).  Costs two extra outer-step gradients,
        one of them in f32 mode; changes no state."""
        cfg = self.config
        img = batch.reshape(batch.shape[0], -1, batch.shape[-1])
        if masks is None:
            g = torch.Generator().manual_seed(0)
            masks = make_masks(self.coords.shape[0], cfg.training.max_num_sampled_points, cfg.meta.num_inner_steps,
                               generator=g, device=self.coords.device)
        kw = dict(optimize_gaussian_window=getattr(cfg.nef, "optimize_gaussian_window", False), second_order=self.second_order)
        lat0, lrs = self._latents0(state), state.params["meta_sgd_lrs"]
        _, own = meta_gradients(self.nef, state.params["nef"], lat0, lrs, self.coords, img, masks, fd_step=self.fd_step, **kw)
        _, ref = meta_gradients(self.nef.with_precision("f32"), state.params["nef"], lat0, lrs, self.coords, img, masks, **kw)
        rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
        gmax = max(float(t.norm()) for t in ref["nef"])
        out = {}
        for path, a, b in zip(tensor_paths(self.nef.num_layers), own["nef"], ref["nef"]):
            if float(b.norm()) > 1e-6 * gmax:
                out["/".join(path)] = rel(a, b)
        vals = sorted(out.values())
        out["median"], out["max"] = vals[len(vals) // 2], vals[-1]
        for k in ref["autodecoder"]:
            if float(ref["autodecoder"][k].norm()) > 0:
                out["lat0/" + k] = rel(own["autodecoder"][k], ref["autodecoder"][k])
        for k in ref["meta_sgd_lrs"]:
            if float(ref["meta_sgd_lrs"][k].norm()) > 0:
                out["lr/" + k] = rel(own["meta_sgd_lrs"][k], ref["meta_sgd_lrs"][k])
        return out

    # ------------------------------------------------------------------ latent-ODE phase (pde_trainer.py:290-500)
    def _fit_initial_latents(self, state, initial_state, masks=None, initial_state_dp=0.0):
        """inner_loop on the first frame of every trajectory (pde_trainer.py:424-427): fitted latents, leading dim B."""
        cfg = self.config
        img = initial_state.reshape(initial_state.shape[0], -1, initial_state.shape[-1])
        coords = self.coords
        if initial_state_dp > 0:                                                  # pde_trainer.py:139-145
            keep = torch.randperm(coords.shape[0], generator=state.rng)[:int(coords.shape[0] * initial_state_dp)].to(coords.device)
            coords, img = coords[keep], img[:, keep]
        if masks is None:
            masks = make_masks(coords.shape[0], cfg.training.max_num_sampled_points, cfg.meta.num_inner_steps,
                               generator=state.rng, device=coords.device)
        return coords, img, masks

    def rollout(self, ode_params, lat, num_frames, graph=False):
        """Latents of ``num_frames`` frames from the fitted ones: (B, T, Z, .) each (pde_trainer.py:432-441).
        graph=True (inference): every derivative evaluation replays one captured hipGraph (PonitaODEGen.graphed)."""
        cfg = self.config
        n_ori = self.nef.cross_attn_invariant.num_z_ori_dims
        z0 = (_pose(lat, n_ori), lat["a"], lat.get("gaussian_window"))
        if graph and hasattr(self.ode_model, "graphed") and not torch.is_grad_enabled():
            # one capture per (parameter tensors, latent shapes): validation sweeps many batches with the same parameters
            leaves = _leaves(ode_params)
            key = (tuple(id(t) for t in leaves), tuple(None if v is None else tuple(v.shape) for v in z0))
            hit = getattr(self, "_ode_graph", None)
            if hit is None or hit[0] != key:
                hit = (key, self.ode_model.graphed(ode_params, z0), leaves)      # (leaves kept alive: ids stay unique)
                self._ode_graph = hit
            f = hit[1]
            return solve_latent_ode(lambda z, t: f(z), z0, 0, num_frames - 1, cfg.node.dt, method=cfg.node.method)
        if graph and torch.is_grad_enabled() and hasattr(self.ode_model, "graphed_train") and z0[1].is_cuda:
            # training: one captured (forward, backward) pair per derivative evaluation of the roll-out; ``ode_params`` must
            # be the persistent leaves of _ode_static_leaves (the graphs keep their addresses)
            leaves = _leaves(ode_params)
            n_eval = (num_frames - 1) * (4 if cfg.node.method == "rk4" else 1)
            key = (tuple(id(t) for t in leaves), tuple(None if v is None else tuple(v.shape) for v in z0), n_eval)
            cache = self.__dict__.setdefault("_ode_train_graphs", {})
            if key not in cache:
                if len(cache) >= 4:
                    cache.clear()
                cache[key] = (self.ode_model.graphed_train(ode_params, z0, n_eval), leaves)
            calls = iter(cache[key][0])
            return solve_latent_ode(lambda z, t: next(calls)(z), z0, 0, num_frames - 1, cfg.node.dt, method=cfg.node.method)
        return solve_latent_ode(lambda z, t: self.ode_model.apply(ode_params, z), z0, 0, num_frames - 1, cfg.node.dt,
                                method=cfg.node.method)

    def _ode_static_leaves(self, ode_params):
        """The ODE parameters as PERSISTENT leaf tensors that require grad, holding the current values: captured training
        evaluations read their parameters by address, the optimiser hands out new tensors every step."""
        cur = _leaves(ode_params)
        st = getattr(self, "_ode_static", None)
        if st is None or len(st) != len(cur) or any(a.shape != b.shape or a.device != b.device for a, b in zip(st, cur)):
            st = [t.detach().clone().requires_grad_(True) for t in cur]
            self._ode_static = st
            self.__dict__.pop("_ode_train_graphs", None)
        else:
            with torch.no_grad():
                torch._foreach_copy_(st, [t.detach() for t in cur])
        return st

    def _ode_train_leaves(self, ode_params):
        if self.graph_ode_training and _leaves(ode_params)[0].is_cuda:
            return self._ode_static_leaves(ode_params), True
        return [t.detach().requires_grad_(True) for t in _leaves(ode_params)], False

    def ode_loss(self, nef_params, ode_params, lat, trajectory, point_masks=None, generator=None, graph=False):
        """pde_trainer.py:411-481 from the fitted latents on: roll the latents out over the training frames, decode every
        frame (at ``max_num_sampled_points`` random grid points per frame when the grid is larger) and compare.
        ``trajectory`` (B, T, *grid, O);  ``point_masks`` (T, n_s) long, or None to draw them."""
        cfg = self.config
        B, T = trajectory.shape[:2]
        sol = self.rollout(ode_params, lat, T, graph=graph)
        p_fl, a_fl, w_fl = (None if v is None else v.reshape(B * T, *v.shape[2:]) for v in sol)
        traj = trajectory.reshape(B, T, -1, trajectory.shape[-1])
        N, n_s = self.coords.shape[0], cfg.training.max_num_sampled_points
        if n_s < N:                                                               # pde_trainer.py:446-471
            if point_masks is None:
                point_masks = torch.stack([torch.randperm(N, generator=generator)[:n_s] for _ in range(T)]).to(self.coords.device)
            xs = self.coords[point_masks][None].expand(B, -1, -1, -1).reshape(B * T, n_s, -1)
            ys = torch.gather(traj, 2, point_masks[None, :, :, None].expand(B, -1, -1, traj.shape[-1])).reshape(B * T, n_s, -1)
        else:
            xs = self.coords[None].expand(B * T, -1, -1)
            ys = traj.reshape(B * T, N, -1)
        recon = self.nef.apply(nef_params, xs, p_fl, a_fl, w_fl)
        return ((recon - ys) ** 2).mean()

    def _fitted(self, state, trajectory, masks):
        coords, img, masks = self._fit_initial_latents(state, trajectory[:, 0], masks)
        cfg = self.config
        _, lat = inner_loop(self.nef, state.params["nef"], self._latents0(state), state.params["meta_sgd_lrs"], coords, img, masks,
                            optimize_gaussian_window=getattr(cfg.nef, "optimize_gaussian_window", False),
                            noise_pos=getattr(cfg.meta, "noise_pos_inner_loop", 0.0), generator=state.rng)
        return {k: v.detach() for k, v in lat.items()}

    def ode_train_step(self, state, trajectory, masks=None, point_masks=None):
        """pde_trainer.py:290-318: one Adam step on the ODE parameters only.  The fitted latents do not depend on them, so
        the inner loop runs without a graph; the gradient flows decoder -> (HIP latent backward) -> solver -> ODE model."""
        cfg = self.config
        trajectory = trajectory[:, :cfg.dataset.traj_len_train]                  # pde_trainer.py:421-422
        lat = self._fitted(state, trajectory, masks)
        leaves, graph = self._ode_train_leaves(state.params["ode_params"])
        ode_params = _unflatten(state.params["ode_params"], leaves)
        loss = self.ode_loss(state.params["nef"], ode_params, lat, trajectory, point_masks, state.rng, graph=graph)
        grads = list(torch.autograd.grad(loss, leaves, allow_unused=True))
        grads = [torch.zeros_like(t) if g is None else g for t, g in zip(leaves, grads)]
        flat = grads + [loss.detach().reshape(1)]
        allreduce_mean_(flat, weight=trajectory.shape[0])
        new_leaves, ode_opt_state = self.ode_opt.update(grads, state.ode_opt_state, [t.detach() for t in leaves])
        params = dict(state.params, ode_params=_unflatten(state.params["ode_params"], new_leaves))
        return flat[-1][0], TrainState(params=params, nef_opt_state=state.nef_opt_state,
                                       autodecoder_opt_state=state.autodecoder_opt_state, meta_sgd_opt_state=state.meta_sgd_opt_state,
                                       ode_opt_state=ode_opt_state, step=state.step + 1, rng=state.rng)

    def dual_train_step(self, state, trajectory, masks=None, point_masks=None):
        """pde_trainer.py:320-358: the roll-out loss trains the nef weights (clip + AdamW), the inner learning rates (Adam,
        clipped) and the ODE parameters (Adam); the latent initialisation is left alone.  The nef / learning-rate
        gradients include the path through the inner loop (the same adjoint recursion as nef_train_step, started from
        d loss / d fitted latents of the roll-out)."""
        cfg = self.config
        trajectory = trajectory[:, :cfg.dataset.traj_len_train]
        coords, img, masks = self._fit_initial_latents(state, trajectory[:, 0], masks)
        leaves, graph = self._ode_train_leaves(state.params["ode_params"])
        ode_params = _unflatten(state.params["ode_params"], leaves)
        if point_masks is None and cfg.training.max_num_sampled_points < self.coords.shape[0]:
            point_masks = torch.stack([torch.randperm(self.coords.shape[0], generator=state.rng)[:cfg.training.max_num_sampled_points]
                                       for _ in range(trajectory.shape[1])]).to(self.coords.device)
        side = {}

        def terminal(weights, lat, keys):
            w = [t.detach().requires_grad_(True) for t in weights]
            lv = {k: lat[k].detach().requires_grad_(True) for k in lat}
            loss = self.ode_loss(_tree_from_tensors(w), ode_params, lv, trajectory, point_masks, graph=graph)
            g = torch.autograd.grad(loss, w + [lv[k] for k in keys] + leaves, allow_unused=True)
            z = lambda t, gi: torch.zeros_like(t) if gi is None else gi
            side["ode"] = [z(t, gi) for t, gi in zip(leaves, g[len(w) + len(keys):])]
            return loss.detach(), [z(t, gi) for t, gi in zip(w, g[:len(w)])], \
                {k: z(lat[k], gi) for k, gi in zip(keys, g[len(w):len(w) + len(keys)])}

        lrs = state.params["meta_sgd_lrs"]
        loss, grads = meta_gradients(self.nef, state.params["nef"], self._latents0(state), lrs, coords, img, masks,
                                     optimize_gaussian_window=getattr(cfg.nef, "optimize_gaussian_window", False),
                                     second_order=self.second_order, fd_step=self.fd_step,
                                     noise_pos=getattr(cfg.meta, "noise_pos_inner_loop", 0.0), generator=state.rng, terminal=terminal)
        lr_keys = list(lrs.keys())
        flat = grads["nef"] + [grads["meta_sgd_lrs"][k] for k in lr_keys] + side["ode"] + [loss.reshape(1)]
        allreduce_mean_(flat, weight=img.shape[0])
        weights = self.nef.param_tensors(state.params["nef"])
        new_w, nef_opt_state = self.nef_opt.update(clip_by_global_norm(grads["nef"], 1.0), state.nef_opt_state, weights)
        new_lrs, lr_state = self.meta_sgd_opt.update([grads["meta_sgd_lrs"][k] for k in lr_keys], state.meta_sgd_opt_state,
                                                     [lrs[k] for k in lr_keys])
        new_leaves, ode_opt_state = self.ode_opt.update(side["ode"], state.ode_opt_state, [t.detach() for t in leaves])
        params = dict(state.params, nef=_tree_from_tensors(new_w), meta_sgd_lrs={k: v.clamp(1e-6, 10.0) for k, v in zip(lr_keys, new_lrs)},
                      ode_params=_unflatten(state.params["ode_params"], new_leaves))
        return flat[-1][0], TrainState(params=params, nef_opt_state=nef_opt_state, autodecoder_opt_state=state.autodecoder_opt_state,
                                       meta_sgd_opt_state=lr_state, ode_opt_state=ode_opt_state, step=state.step + 1, rng=state.rng)

    def select_train_step(self, epoch):
        """The phase schedule of _base_pde_trainer.py:280-299: nef while training.nef.train_from_epoch < epoch <=
        train_until_epoch, ode likewise, both -> dual.  Returns the bound step function; every step takes
        (state, trajectory) -- the nef phase fits the frames nef_loss picks (_nef_frames)."""
        t = self.config.training
        train_nef = t.nef.train_from_epoch < epoch <= t.nef.train_until_epoch
        train_ode = self.ode_model is not None and t.ode.train_from_epoch < epoch <= t.ode.train_until_epoch
        if train_nef and train_ode:
            return self.dual_train_step
        if train_nef:
            return lambda state, trajectory, **kw: self.nef_train_step(state, self._nef_frames(state, trajectory), **kw)
        if train_ode:
            return self.ode_train_step
        raise ValueError("No training step set")

    def _nef_frames(self, state, trajectory):
        """nef_loss's choice of frames (pde_trainer.py:483-497): the first frame, or ``fit_on_num_steps`` random training
        frames of every trajectory, each fitted as a signal of its own."""
        k = getattr(getattr(self.config.training, "nef", None), "fit_on_num_steps", 1)
        if k == 1:
            return trajectory[:, 0]
        idx = torch.randperm(self.config.dataset.traj_len_train, generator=state.rng)[:k].to(trajectory.device)
        sub = trajectory[:, idx]
        return sub.reshape(sub.shape[0] * sub.shape[1], *sub.shape[2:])

    def train_epoch(self, state, loader, epoch):
        """One pass over ``loader`` (an iterable of trajectories (B, T, *grid, O) or of the reference's
        (trajectory, _, _) batches) with the step the schedule selects; returns (mean loss, state)."""
        step = self.select_train_step(epoch)
        total, n = 0.0, 0
        for batch in loader:
            trajectory = batch[0] if isinstance(batch, (tuple, list)) else batch
            loss, state = step(state, trajectory)
            total, n = total + float(loss), n + 1
        return total / max(n, 1), state

    @torch.no_grad()
    def val_step(self, state, trajectory, initial_state_dp=0.0, masks=None):
        """pde_trainer.py:360-409: fit the first frame, roll out over train + out-of-horizon frames, decode the full grid;
        returns (mse over the training horizon, mse beyond it)."""
        cfg = self.config
        T_in = cfg.dataset.traj_len_train
        trajectory = trajectory[:, :T_in + cfg.dataset.traj_len_out_horizon]
        B, T = trajectory.shape[:2]
        coords, img, masks = self._fit_initial_latents(state, trajectory[:, 0], masks, initial_state_dp)
        with torch.enable_grad():
            _, lat = inner_loop(self.nef, state.params["nef"], self._latents0(state), state.params["meta_sgd_lrs"], coords, img, masks,
                                optimize_gaussian_window=getattr(cfg.nef, "optimize_gaussian_window", False))
        sol = self.rollout(state.params["ode_params"], {k: v.detach() for k, v in lat.items()}, T, graph=T > 4)
        p_fl, a_fl, w_fl = (None if v is None else v.reshape(B * T, *v.shape[2:]) for v in sol)
        recon = decode(self.nef, state.params["nef"], self.coords, p_fl, a_fl, w_fl).reshape(trajectory.shape)
        err = (recon - trajectory) ** 2
        return err[:, :T_in].mean(), (err[:, T_in:].mean() if T > T_in else err.new_zeros(()))
