"""Auto-decoder (non-meta) ENF trainer -- the nef phase of experiments/fitting/trainers/nonmaml_pde_trainer.py.

Every training signal owns a row of latents in a PositionOrientationFeatureAutodecoder (:37-45); one step is

    recon_loss, grads = jax.value_and_grad(self.enf_loss)(params, state, autodecoder_fn, trajectory, mask, traj_idx)   (:118)
    nef:          clip_by_global_norm(1.0) -> adamw(lr_enf)                                                        (:63-66,121-122)
    autodecoder:  adam(lr_codes) over the WHOLE latent table (rows outside the batch move by momentum only)          (:67,125-126)

with enf_loss = mean((nef.apply(params['nef'], coords[mask], *autodecoder(params['autodecoder'], traj_idx)) - state)^2)
(:309-341).  The gradient is first order, so it is exactly what the training path of the decoder provides (weight
gradients through the HIP pair kernels' activation store, latent gradients through the latent table).
"""
from dataclasses import dataclass, field

import torch

from ..optim import Adam, AdamW, clip_by_global_norm
from ..parallel import allreduce_mean_
from .pde_trainer import _tree_from_tensors


@dataclass
class NonMetaTrainState:
    params: dict
    nef_opt_state: dict
    autodecoder_opt_state: dict
    step: int = 0
    rng: torch.Generator = field(default_factory=lambda: torch.Generator().manual_seed(0))


class NonMetaPDETrainer:
    """``config`` fields used: optimizer.learning_rate_enf, optimizer.learning_rate_codes,
    training.max_num_sampled_points.  ``autodecoder``: enf_pde_amd.enf.latents.autodecoder.PositionOrientationFeatureAutodecoder
    sized for the training set.

    Scope: SURVEY.md 8f row 1 asks for the nef phase (the first-order decoder gradients this build accelerates).  The
    reference's other two steps of this trainer -- ``_ode_train_step`` (:173-199, clip + adamw on the latent ODE over the
    auto-decoder's latents) and ``_val_step`` (:201-241, fit validation latents from scratch, then roll out) -- are beyond
    section 8; they exist here as methods that raise, so that a config which schedules them fails loudly instead of
    silently training less than the reference does.  The MAML trainer (pde_trainer.py) has both phases."""

    def __init__(self, config, nef, autodecoder, coords, seed=42):
        self.config, self.nef, self.autodecoder, self.coords, self.seed = config, nef, autodecoder, coords, seed
        self.nef_opt = AdamW(config.optimizer.learning_rate_enf)
        self.autodecoder_opt = Adam(config.optimizer.learning_rate_codes)

    def init_train_state(self, nef_params=None):
        dev = self.coords.device
        g = torch.Generator().manual_seed(self.seed)
        ad = self.autodecoder.init(g, device=dev)
        if nef_params is None:
            nef_params = self.nef.init(g, device=dev)
        return NonMetaTrainState(params={"nef": nef_params, "autodecoder": ad},
                                 nef_opt_state=self.nef_opt.init(self.nef.param_tensors(nef_params)),
                                 autodecoder_opt_state=self.autodecoder_opt.init(list(ad["params"].values())), step=0, rng=g)

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

    def loss_and_grads(self, state, initial_state, traj_idx, mask=None):
        """(recon_loss, grads['nef'] as 46 tensors, grads['autodecoder'] as dense tensors like the latent table)."""
        cfg = self.config
        img = initial_state.reshape(initial_state.shape[0], -1, initial_state.shape[-1])
        coords = self.coords
        if mask is not None:                                                              # :321-323
            img, coords = img[:, mask], coords[mask]
        npts = cfg.training.max_num_sampled_points
        if npts < coords.shape[0]:                                                        # :326-335
            sub = torch.randperm(coords.shape[0], generator=state.rng)[:npts].to(coords.device)
            img, coords = img[:, sub], coords[sub]
        P = state.params["autodecoder"]["params"]
        names = list(P.keys())
        leaves = {k: P[k].detach().requires_grad_(True) for k in names}
        w = [t.detach().requires_grad_(True) for t in self.nef.param_tensors(state.params["nef"])]
        p, a, window = self.autodecoder.apply({"params": leaves}, traj_idx)               # :338
        xs = coords[None].expand(img.shape[0], -1, -1)
        out = self.nef.apply(_tree_from_tensors(w), xs, p, a, window)                     # :341
        loss = ((out - img) ** 2).mean()
        g = torch.autograd.grad(loss, w + [leaves[k] for k in names], allow_unused=True)
        gw = [torch.zeros_like(t) if gi is None else gi for t, gi in zip(w, g[:len(w)])]
        ga = [torch.zeros_like(leaves[k]) if gi is None else gi for k, gi in zip(names, g[len(w):])]
        return loss.detach(), gw, dict(zip(names, ga))

    def _step(self, state, batch, mask, update_nef):
        initial_state, traj_idx = batch
        loss, gw, ga = self.loss_and_grads(state, initial_state, traj_idx, mask)
        names = list(ga.keys())
        flat = gw + [ga[k] for k in names] + [loss.reshape(1)]
        allreduce_mean_(flat, weight=initial_state.shape[0])
        loss = flat[-1][0]
        nef_params, nef_opt_state = state.params["nef"], state.nef_opt_state
        if update_nef:
            new_w, nef_opt_state = self.nef_opt.update(clip_by_global_norm(gw, 1.0), state.nef_opt_state,
                                                       self.nef.param_tensors(state.params["nef"]))
            nef_params = _tree_from_tensors(new_w)
        P = state.params["autodecoder"]["params"]
        new_p, ad_state = self.autodecoder_opt.update([ga[k] for k in names], state.autodecoder_opt_state, [P[k] for k in names])
        return loss, NonMetaTrainState(params={"nef": nef_params, "autodecoder": {"params": dict(zip(names, new_p))}},
                                       nef_opt_state=nef_opt_state, autodecoder_opt_state=ad_state, step=state.step + 1,
                                       rng=state.rng)

    def nef_train_step(self, state, batch, mask=None):
        """batch = (initial states (B, ..., O), trajectory indices (B,) long)   (:101-137)"""
        return self._step(state, batch, mask, True)

    def nef_train_step_autodec_only(self, state, batch, mask=None):
        """Only the latents move (:139-171)."""
        return self._step(state, batch, mask, False)

    def ode_train_step(self, state, batch):
        raise NotImplementedError("NonMetaPDETrainer: the latent-ODE phase of nonmaml_pde_trainer.py:173-199 is outside "
                                  "SURVEY.md section 8; use MetaSGDPDETrainer.ode_train_step for the latent ODE")

    def val_step(self, state, batch):
        raise NotImplementedError("NonMetaPDETrainer: the validation roll-out of nonmaml_pde_trainer.py:201-241 is outside "
                                  "SURVEY.md section 8; use MetaSGDPDETrainer.val_step")
