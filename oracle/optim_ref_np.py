"""ORACLE (test infrastructure, not product code) -- numpy fp64 restatement of the optax update rules the
reference's outer loop uses (experiments/fitting/trainers/pde_trainer.py:60-67,258-273).

optax is a third-party dependency absent from /root/reference and from this image; its version is un-pinned
(README.md:31: `pip install "jax[cuda12]" flax optax orbax`).  PARITY UNPINNED: the reference holds no
fixture for an optimiser step; the rules restated here are optax's documented ones:
  scale_by_adam:       mu = b1 mu + (1-b1) g ; nu = b2 nu + (1-b2) g^2 ; count += 1
                       u = (mu / (1 - b1^count)) / (sqrt(nu / (1 - b2^count) + eps_root) + eps),  eps_root = 0
  adam(lr):            p <- p - lr u
  adamw(lr, wd=1e-4):  p <- p - lr (u + wd p)
  clip_by_global_norm(c): g <- g / max(1, ||g||_2 / c)  over the whole tree
Only tests/ may import this module.
"""
import numpy as np


def clip_by_global_norm(grads, c=1.0):
    n = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return [g / max(1.0, n / c) for g in grads]


def adam_step(params, grads, state, lr, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0):
    count = state["count"] + 1
    mu = [b1 * m + (1 - b1) * g for m, g in zip(state["mu"], grads)]
    nu = [b2 * v + (1 - b2) * g * g for v, g in zip(state["nu"], grads)]
    out = []
    for p, m, v in zip(params, mu, nu):
        u = (m / (1 - b1 ** count)) / (np.sqrt(v / (1 - b2 ** count)) + eps)
        out.append(p - lr * (u + weight_decay * p))
    return out, {"count": count, "mu": mu, "nu": nu}


def init_state(params):
    return {"count": 0, "mu": [np.zeros_like(p) for p in params], "nu": [np.zeros_like(p) for p in params]}
