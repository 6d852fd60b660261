"""CPU tests of the latent-ODE oracle (oracle/ode_ref_np.py, ode_ref_torch.py): the two restatements agree, the
model is permutation-equivariant over latents and equivariant to the symmetry its invariant encodes, gradients
match finite differences, and the solvers integrate a linear system as Euler / RK4 must."""
import numpy as np
import pytest
import torch

from oracle import enf_ref_torch as T
from oracle import ode_ref_np as O
from oracle import ode_ref_torch as OT

SA_INV = ["rel_pos_periodic", "ponita", "polar_periodic", "latitude_periodic", "rel_pos", "norm_rel_pos", "abs_pos", "ball"]


def ode_cfg(inv, **kw):
    d = dict(invariant=inv, num_in=3 if inv in ("ball", "ball_lat") else 2, num_hidden=16, num_layers=2, basis_dim=8,
             degree=3, widening_factor=2, kernel_size="global", vec_num_out=1, global_pool=False)
    d.update(kw)
    return d


def ode_inputs(cfg, B, Z, C, seed):
    rng = np.random.default_rng(seed)
    spec = O.sa_invariant_spec(cfg["invariant"], cfg["num_in"])
    if cfg["invariant"] in ("polar_periodic", "latitude_periodic"):
        p = np.stack([rng.uniform(0, 2 * np.pi, (B, Z)), rng.uniform(0.3, np.pi - 0.3, (B, Z))], -1)
    elif cfg["invariant"] in ("ball", "ball_lat"):
        p = np.stack([rng.uniform(0, 2 * np.pi, (B, Z)), rng.uniform(0.3, np.pi - 0.3, (B, Z)),
                      rng.uniform(0, 2 * np.pi, (B, Z)), rng.uniform(0.5, 1.0, (B, Z))], -1)
    else:
        p = rng.uniform(-1, 1, (B, Z, spec["z_pos"]))
        if spec["z_ori"]:
            p = np.concatenate([p, rng.uniform(-np.pi, np.pi, (B, Z, 1))], -1)
    return p, 1 + 0.3 * rng.standard_normal((B, Z, C)), np.full((B, Z, 1), 0.5)


def test_poly_feature_count_and_order():
    assert O.num_poly_features(4, 3) == 340 and O.num_poly_features(3, 3) == 120 and O.num_poly_features(5, 3) == 780
    x = np.array([[2.0, 3.0]])
    f = O.poly_features(x, 2)[0]                                    # [x | x(x)x | (x(x)x)(x)x]
    assert np.array_equal(f, [2, 3, 4, 6, 6, 9, 8, 12, 12, 18, 12, 18, 18, 27])


@pytest.mark.parametrize("inv", SA_INV)
def test_numpy_and_torch_restatements_agree(inv):
    cfg = ode_cfg(inv)
    prm = O.init_ponita_ode(1, cfg, latent_dim=5, jitter=0.1, readout_scale=1.0)
    lat = ode_inputs(cfg, 2, 6, 5, 2)
    dp, da, dw = O.ponita_ode(prm, cfg, lat)
    tp, ta, tw = OT.ponita_ode(T.to_torch(prm, torch.float64), cfg, tuple(torch.tensor(v) for v in lat))
    assert dp.shape == lat[0].shape and da.shape == lat[1].shape and not dw.any()
    assert np.abs(tp.numpy() - dp).max() < 1e-11 and np.abs(ta.numpy() - da).max() < 1e-11 and not tw.any()
    assert np.abs(dp).max() > 1e-3 and np.abs(da).max() > 1e-3


def test_kernel_size_window_and_global_pool():
    cfg = ode_cfg("ponita", kernel_size=0.2)                        # config_cahn_hilliard.yaml
    prm = O.init_ponita_ode(3, cfg, latent_dim=4, readout_scale=1.0)
    lat = ode_inputs(cfg, 2, 5, 4, 4)
    dp, da, _ = O.ponita_ode(prm, cfg, lat)
    tp, ta, _ = OT.ponita_ode(T.to_torch(prm, torch.float64), cfg, tuple(torch.tensor(v) for v in lat))
    assert np.abs(tp.numpy() - dp).max() < 1e-11 and np.abs(ta.numpy() - da).max() < 1e-11
    g = dict(cfg, global_pool=True)
    sc, vec = O.ponita_gen(prm["params"]["ponita"], g, lat[0], lat[1] - 1)
    assert sc.shape == (2, 5) and vec.shape == (2, 2)               # pooled over latents (PODE:189-193)


def test_latent_permutation_equivariance():
    cfg = ode_cfg("rel_pos_periodic")
    prm = O.init_ponita_ode(5, cfg, latent_dim=4, jitter=0.1, readout_scale=1.0)
    p, a, w = ode_inputs(cfg, 2, 7, 4, 6)
    perm = np.random.default_rng(0).permutation(7)
    dp, da, _ = O.ponita_ode(prm, cfg, (p, a, w))
    dp2, da2, _ = O.ponita_ode(prm, cfg, (p[:, perm], a[:, perm], w[:, perm]))
    assert np.abs(dp[:, perm] - dp2).max() < 1e-12 and np.abs(da[:, perm] - da2).max() < 1e-12


@pytest.mark.parametrize("inv", ["rel_pos", "rel_pos_periodic", "norm_rel_pos"])
def test_translation_invariance(inv):
    cfg = ode_cfg(inv)
    prm = O.init_ponita_ode(7, cfg, latent_dim=4, readout_scale=1.0)
    p, a, w = ode_inputs(cfg, 2, 6, 4, 8)
    t = np.array([0.37, -0.21])
    dp, da, _ = O.ponita_ode(prm, cfg, (p, a, w))
    dp2, da2, _ = O.ponita_ode(prm, cfg, (p + t, a, w))
    assert np.abs(dp - dp2).max() < 1e-10 and np.abs(da - da2).max() < 1e-10


def test_ponita_se2_equivariance():
    """Rotating and translating every pose leaves da/dt and the angle rate unchanged and rotates dp_pos/dt."""
    cfg = ode_cfg("ponita")
    prm = O.init_ponita_ode(9, cfg, latent_dim=4, readout_scale=1.0)
    p, a, w = ode_inputs(cfg, 2, 6, 4, 10)
    th, t = 0.7, np.array([0.3, -0.4])
    Q = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    p2 = np.concatenate([p[..., :2] @ Q.T + t, p[..., 2:] + th], -1)
    dp, da, _ = O.ponita_ode(prm, cfg, (p, a, w))
    dp2, da2, _ = O.ponita_ode(prm, cfg, (p2, a, w))
    assert np.abs(da - da2).max() < 1e-10 and np.abs(dp[..., 2] - dp2[..., 2]).max() < 1e-10
    assert np.abs(dp[..., :2] @ Q.T - dp2[..., :2]).max() < 1e-10


@pytest.mark.parametrize("inv", ["rel_pos_periodic", "ponita", "latitude_periodic"])
def test_gradients_match_finite_differences(inv):
    cfg = ode_cfg(inv, num_hidden=8, basis_dim=4, num_layers=1)
    prm = T.to_torch(O.init_ponita_ode(11, cfg, latent_dim=3, readout_scale=1.0), torch.float64)
    p, a, w = (torch.tensor(v) for v in ode_inputs(cfg, 1, 4, 3, 12))
    rng = np.random.default_rng(1)
    wp, wa = torch.tensor(rng.standard_normal(p.shape)), torch.tensor(rng.standard_normal(a.shape))

    def f(p_, a_):
        dp, da, _ = OT.ponita_ode(prm, cfg, (p_, a_, w))
        return (dp * wp).sum() + (da * wa).sum()
    assert torch.autograd.gradcheck(f, (p.clone().requires_grad_(True), a.clone().requires_grad_(True)),
                                    eps=1e-6, atol=1e-6, rtol=1e-4)


def test_mlp_ode_restatements_agree():
    prm = O.init_mlp_ode(13, 16, 2, 5)
    rng = np.random.default_rng(14)
    lat = (rng.uniform(-1, 1, (2, 6, 2)), 1 + 0.2 * rng.standard_normal((2, 6, 5)), np.ones((2, 6, 1)))
    dp, da, dw = O.mlp_ode(prm, lat)
    tp, ta, _ = OT.mlp_ode(T.to_torch(prm, torch.float64), tuple(torch.tensor(v) for v in lat))
    assert dp.shape == (2, 6, 2) and da.shape == (2, 6, 5) and not dw.any()
    assert np.abs(tp.numpy() - dp).max() < 1e-12 and np.abs(ta.numpy() - da).max() < 1e-12


def test_solvers_on_a_linear_system():
    f = lambda x, t: (-x[0], 2.0 * x[1], np.zeros_like(x[2]))
    x0 = (np.ones((2, 3, 2)), np.ones((2, 3, 4)), np.full((2, 3, 1), 0.7))
    pe, ae, we = O.solve_latent_ode(f, x0, 0, 4, 0.5, method="euler")
    assert pe.shape == (2, 9, 3, 2) and ae.shape == (2, 9, 3, 4) and we.shape == (2, 9, 3, 1)
    k = np.arange(9)
    assert np.allclose(pe[0, :, 0, 0], 0.5 ** k) and np.allclose(ae[0, :, 0, 0], 2.0 ** k) and np.allclose(we, 0.7)
    pr, ar, _ = O.solve_latent_ode(f, x0, 0, 4, 0.5, method="rk4")
    g = 1 - 0.5 + 0.5 ** 2 / 2 - 0.5 ** 3 / 6 + 0.5 ** 4 / 24                     # RK4's amplification factor
    assert np.allclose(pr[0, :, 0, 0], g ** k) and abs(pr[0, -1, 0, 0] - np.exp(-4)) < 2e-3
    with pytest.raises(ValueError, match="Unknown method"):
        O.solve_latent_ode(f, x0, 0, 1, 0.5, method="heun")
    ft = lambda x, t: (-x[0], 2.0 * x[1], torch.zeros_like(x[2]))
    tr = OT.solve_latent_ode(ft, tuple(torch.tensor(v) for v in x0), 0, 4, 0.5, method="rk4")
    assert np.allclose(tr[0].numpy(), pr) and np.allclose(tr[1].numpy(), ar)


@pytest.mark.parametrize("name", ["ode_rel_pos_periodic", "ode_ponita"])
def test_oracle_reproduces_ode_golden(name):
    import os
    from tests.golden.make_golden import make_ode_case
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    _, rec = make_ode_case(name)
    for k in g.files:
        assert np.allclose(np.asarray(rec[k]), g[k], rtol=1e-9, atol=1e-12), (name, k)
