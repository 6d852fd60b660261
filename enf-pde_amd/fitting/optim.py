"""Outer-loop optimisers with optax's update rules (the reference builds them in
experiments/fitting/trainers/pde_trainer.py:60-67):

    nef_opt          = optax.chain(optax.clip_by_global_norm(1.0), optax.adamw(lr_enf))
    autodecoder_opt  = optax.adam(lr_codes)
    meta_sgd_opt     = optax.adam(lr_meta_sgd)          # followed by clip(lrs, 1e-6, 10), pde_trainer.py:272

optax is not installed here and its version is un-pinned in the reference (README.md:31); the rules below are
optax's published ones (scale_by_adam with bias correction, eps outside the square root, eps_root = 0;
adamw = scale_by_adam -> add_decayed_weights(1e-4) -> scale(-lr); clip_by_global_norm: g / max(1, ||g|| / c)).
Functional style like optax: ``init(params) -> state``, ``update(grads, state, params) -> (new_params, state)``;
trees are flat lists of tensors.  Pure tensor code, device-agnostic (tested on CPU against oracle/optim_ref_np.py).
"""
import torch


def global_norm(tensors):
    norms = torch._foreach_norm([t.detach() for t in tensors])
    return torch.linalg.vector_norm(torch.stack(norms))


def clip_by_global_norm(grads, max_norm=1.0):
    """optax.clip_by_global_norm: every leaf scaled by 1 / max(1, ||g||_2 / max_norm)."""
    scale = 1.0 / torch.clamp(global_norm(grads) / max_norm, min=1.0)
    return list(torch._foreach_mul(list(grads), scale))


class Adam:
    """optax.adam(lr, b1=0.9, b2=0.999, eps=1e-8); weight_decay > 0 gives optax.adamw (decoupled, default 1e-4)."""

    def __init__(self, lr, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0):
        self.lr, self.b1, self.b2, self.eps, self.wd = float(lr), b1, b2, eps, weight_decay

    def init(self, params):
        return {"count": 0, "mu": [torch.zeros_like(p) for p in params], "nu": [torch.zeros_like(p) for p in params]}

    @torch.no_grad()
    def update(self, grads, state, params):
        # multi-tensor (foreach) arithmetic: a handful of launches for the whole tree instead of ~10 per leaf
        count = state["count"] + 1
        grads = [g.to(p.dtype) for g, p in zip(grads, params)]
        mu = torch._foreach_mul(state["mu"], self.b1)
        torch._foreach_add_(mu, grads, alpha=1 - self.b1)
        nu = torch._foreach_mul(state["nu"], self.b2)
        torch._foreach_addcmul_(nu, grads, grads, value=1 - self.b2)
        c1, c2 = 1 - self.b1 ** count, 1 - self.b2 ** count
        den = torch._foreach_div(nu, c2)
        torch._foreach_sqrt_(den)
        torch._foreach_add_(den, self.eps)
        upd = torch._foreach_div(mu, den)
        torch._foreach_div_(upd, c1)
        if self.wd:
            torch._foreach_add_(upd, params, alpha=self.wd)
        new = torch._foreach_add(params, upd, alpha=-self.lr)
        return list(new), {"count": count, "mu": list(mu), "nu": list(nu)}


def AdamW(lr, weight_decay=1e-4, **kw):
    return Adam(lr, weight_decay=weight_decay, **kw)
