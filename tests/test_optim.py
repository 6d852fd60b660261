"""Outer-loop optimiser rules (host logic, CPU): enf_pde_amd.fitting.optim against the numpy restatement of the
optax rules in oracle/optim_ref_np.py (pde_trainer.py:60-67,258-273)."""
import numpy as np
import torch

from oracle import optim_ref_np as O
from enf_pde_amd.fitting.optim import Adam, AdamW, clip_by_global_norm


def _rand_tree(rng, shapes):
    return [rng.standard_normal(s) for s in shapes]


def test_clip_by_global_norm():
    rng = np.random.default_rng(0)
    shapes = [(5, 3), (7,), (2, 2, 2)]
    for scale in (0.01, 10.0):
        g = [scale * t for t in _rand_tree(rng, shapes)]
        ref = O.clip_by_global_norm(g, 1.0)
        got = clip_by_global_norm([torch.tensor(t) for t in g], 1.0)
        for a, b in zip(got, ref):
            np.testing.assert_allclose(a.numpy(), b, rtol=1e-12)
    # below the threshold nothing changes; above it the norm becomes exactly 1
    big = clip_by_global_norm([torch.tensor(10.0 * t) for t in _rand_tree(rng, shapes)], 1.0)
    assert abs(float(torch.sqrt(sum((t ** 2).sum() for t in big))) - 1.0) < 1e-12


def _run(opt, ref_kw, steps=5):
    rng = np.random.default_rng(1)
    shapes = [(4, 6), (6,), (3, 1, 2)]
    p_np = _rand_tree(rng, shapes)
    p_t = [torch.tensor(t) for t in p_np]
    st_np, st_t = O.init_state(p_np), opt.init(p_t)
    for _ in range(steps):
        g = _rand_tree(rng, shapes)
        g[1] = np.zeros_like(g[1])          # a gradient-less leaf (the frozen RFF coefficients under AdamW)
        p_np, st_np = O.adam_step(p_np, g, st_np, **ref_kw)
        p_t, st_t = opt.update([torch.tensor(t) for t in g], st_t, p_t)
        for a, b in zip(p_t, p_np):
            np.testing.assert_allclose(a.numpy(), b, rtol=1e-10, atol=1e-14)
    return p_t


def test_adam_matches_reference_rule():
    _run(Adam(3e-3), dict(lr=3e-3))


def test_adamw_decays_gradientless_leaves():
    p0 = np.random.default_rng(1).standard_normal((6,))
    p = _run(AdamW(1e-2), dict(lr=1e-2, weight_decay=1e-4), steps=3)
    # zero gradient -> the Adam term is 0 / (0 + eps) = 0, only the decoupled decay acts: p <- p (1 - lr wd) per step
    del p0
    assert p[1].abs().max() > 0


def test_trainer_imports_and_config_surface():
    from enf_pde_amd.fitting import MetaSGDPDETrainer, meta_gradients  # noqa: F401
