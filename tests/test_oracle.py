"""CPU tests of the oracle: two independent restatements agree, the group-invariance properties the
reference only eyeballs (trainers/_base_pde_trainer.py:731-757) hold numerically, gradients match
finite differences, and the committed golden vectors are reproduced."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from oracle import enf_ref_torch as T
from tests.helpers import make_cfg, make_inputs
from tests.golden.make_golden import CASES

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ALL_INV = ["rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos", "rel_pos", "norm_rel_pos",
           "ball", "ball_lat"]


def small_cfg(inv, **kw):
    d = dict(invariant=inv, D=32, H=2, C=6, O=2, freq=(0.3, 0.7))
    d.update(kw)
    return make_cfg(**d)


def test_param_count_matches_survey():
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    assert R.count_params(R.init_params(0, cfg)) == 531585


@pytest.mark.parametrize("inv", ALL_INV)
def test_numpy_and_torch_restatements_agree(inv):
    cfg = small_cfg(inv)
    prm = R.init_params(1, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 13, 5, 2)
    ref = R.nef_apply(prm, cfg, x, p, a, s)
    o64 = T.nef_apply(T.to_torch(prm, torch.float64), cfg, *(torch.tensor(v) for v in (x, p, a, s))).numpy()
    assert np.abs(o64 - ref).max() < 1e-11
    o32 = T.nef_apply(T.to_torch(prm, torch.float32), cfg, *(torch.tensor(v, dtype=torch.float32) for v in (x, p, a, s))).numpy()
    assert np.abs(o32 - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_chunked_decode_equals_full():
    cfg = small_cfg("rel_pos_periodic")
    prm = T.to_torch(R.init_params(3, cfg), torch.float64)
    x, p, a, s = (torch.tensor(v) for v in make_inputs(cfg, 2, 37, 4, 4))
    assert torch.allclose(T.nef_apply_chunked(prm, cfg, x, p, a, s, chunk=8), T.nef_apply(prm, cfg, x, p, a, s), atol=1e-12)


def _apply(cfg, prm, x, p, a, s):
    return R.nef_apply(prm, cfg, x, p, a, s)


def test_latent_permutation_invariance():
    cfg = small_cfg("rel_pos_periodic")
    prm = R.init_params(5, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 9, 6, 6)
    perm = np.random.default_rng(0).permutation(6)
    assert np.abs(_apply(cfg, prm, x, p, a, s) - _apply(cfg, prm, x, p[:, perm], a[:, perm], s[:, perm])).max() < 1e-12


@pytest.mark.parametrize("inv", ["rel_pos", "rel_pos_periodic", "norm_rel_pos"])
def test_translation_invariance(inv):
    cfg = small_cfg(inv)
    prm = R.init_params(7, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 9, 5, 8)
    t = np.array([0.37, -0.21])
    assert np.abs(_apply(cfg, prm, x, p, a, s) - _apply(cfg, prm, x + t, p + t, a, s)).max() < 1e-10


def test_periodic_shift_by_two():
    cfg = small_cfg("rel_pos_periodic")
    prm = R.init_params(9, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 1, 9, 5, 10)
    x2, p2 = x.copy(), p.copy()
    x2[..., 0] += 2.0
    p2[:, 2, 1] -= 2.0
    assert np.abs(_apply(cfg, prm, x, p, a, s) - _apply(cfg, prm, x2, p2, a, s)).max() < 1e-10


def test_abs_pos_is_not_translation_invariant():
    cfg = small_cfg("abs_pos")
    prm = R.init_params(11, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 1, 9, 5, 12)
    assert np.abs(_apply(cfg, prm, x, p, a, s) - _apply(cfg, prm, x + 0.4, p + 0.4, a, s)).max() > 1e-6


def test_ponita_se2_invariance():
    """Joint roto-translation of queries and latent positions, latent angle shifted with it."""
    cfg = small_cfg("ponita")
    prm = R.init_params(13, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 11, 5, 14)
    al, t = 0.7, np.array([0.3, -0.5])
    Rm = np.array([[np.cos(al), -np.sin(al)], [np.sin(al), np.cos(al)]])
    x2 = x @ Rm.T + t
    p2 = p.copy()
    p2[..., :2] = p[..., :2] @ Rm.T + t
    p2[..., 2] = p[..., 2] + al
    assert np.abs(_apply(cfg, prm, x, p, a, s) - _apply(cfg, prm, x2, p2, a, s)).max() < 1e-10


def _rot_sphere(ang, Q):
    v = np.stack([np.sin(ang[..., 1]) * np.cos(ang[..., 0]), np.sin(ang[..., 1]) * np.sin(ang[..., 0]), np.cos(ang[..., 1])], -1) @ Q.T
    return np.stack([np.mod(np.arctan2(v[..., 1], v[..., 0]), 2 * np.pi), np.arccos(np.clip(v[..., 2], -1, 1))], -1)


def test_polar_periodic_so3_invariance():
    cfg = small_cfg("polar_periodic")
    prm = R.init_params(15, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, 11, 5, 16)
    Q, _ = np.linalg.qr(np.random.default_rng(3).standard_normal((3, 3)))
    Q *= np.sign(np.linalg.det(Q))
    assert np.abs(_apply(cfg, prm, x, p, a, s) - _apply(cfg, prm, _rot_sphere(x, Q), _rot_sphere(p, Q), a, s)).max() < 1e-8


def test_latitude_periodic_longitude_shift_only():
    cfg = small_cfg("latitude_periodic")
    prm = R.init_params(17, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 1, 11, 5, 18)
    sh = np.array([0.9, 0.0])
    base = _apply(cfg, prm, x, p, a, s)
    assert np.abs(base - _apply(cfg, prm, x + sh, p + sh, a, s)).max() < 1e-9
    tilt = np.array([0.0, 0.2])
    assert np.abs(base - _apply(cfg, prm, x + tilt, p + tilt, a, s)).max() > 1e-6


def test_ball_invariants_structure():
    """ball.py:54-96 / ball_lat.py:66-88: R(alpha, beta, gamma) is a rotation; the first three ball invariants are the
    rotated unit vector of the query; ball_lat only sees the longitude DIFFERENCE."""
    cfg = small_cfg("ball")
    x, p, a, s = make_inputs(cfg, 2, 11, 5, 30)
    Rm = R.ball_rotation(p)
    assert np.abs(Rm @ np.swapaxes(Rm, -1, -2) - np.eye(3)).max() < 1e-12 and np.allclose(np.linalg.det(Rm), 1.0)
    inv = R.invariant("ball", x, p)
    assert inv.shape == (2, 11, 5, 5) and np.abs(np.linalg.norm(inv[..., :3], axis=-1) - 1).max() < 1e-12
    assert np.array_equal(inv[..., 3], np.broadcast_to(x[:, :, None, 2], inv.shape[:3]))
    assert np.array_equal(inv[..., 4], np.broadcast_to(p[:, None, :, 3], inv.shape[:3]))
    cfg = small_cfg("ball_lat")
    prm = R.init_params(31, cfg, jitter=0.1)
    base = _apply(cfg, prm, x, p, a, s)
    sx, sp = np.array([0.9, 0.0, 0.0]), np.array([0.9, 0.0, 0.0, 0.0])
    assert np.abs(base - _apply(cfg, prm, x + sx, p + sp, a, s)).max() < 1e-9
    tx, tp = np.array([0.0, 0.2, 0.0]), np.array([0.0, 0.2, 0.0, 0.0])
    assert np.abs(base - _apply(cfg, prm, x + tx, p + tp, a, s)).max() > 1e-6


@pytest.mark.parametrize("inv", ["rel_pos_periodic", "ponita", "polar_periodic", "latitude_periodic", "ball", "ball_lat"])
def test_latent_gradients_match_finite_differences(inv):
    cfg = small_cfg(inv, D=16, O=1)
    prm = T.to_torch(R.init_params(19, cfg, jitter=0.1), torch.float64)
    x, p, a, s = (torch.tensor(v) for v in make_inputs(cfg, 1, 6, 3, 20))
    w = torch.tensor(np.random.default_rng(4).standard_normal((1, 6, 1)))

    def f(p_, a_, s_):
        return (T.nef_apply(prm, cfg, x, p_, a_, s_) * w).sum()
    assert torch.autograd.gradcheck(f, (p.clone().requires_grad_(True), a.clone().requires_grad_(True),
                                        s.clone().requires_grad_(True)), eps=1e-6, atol=1e-6, rtol=1e-4)


def test_latent_init_rules():
    lat = R.init_latents(3, 64, 16, "rel_pos_periodic")
    assert lat["p_pos"].shape == (3, 64, 2) and np.allclose(lat["gaussian_window"], 2 / 8)     # AD:43
    ax = np.linspace(-1 + 1 / 8, 1 - 1 / 8, 8)
    assert np.allclose(lat["p_pos"][0, :8, 1], ax) and np.allclose(lat["p_pos"][0, ::8, 0], ax)  # 'ij' order, LU:95
    assert np.all(lat["a"] == 1.0)
    pol = R.init_latents(1, 128, 32, "latitude_periodic", coordinate_system="polar")
    assert pol["p_pos"].shape == (1, 128, 2) and np.allclose(pol["gaussian_window"], 2 * np.pi / 8)  # AD:51
    pon = R.init_latents(1, 16, 8, "ponita")
    assert pon["p_ori"].shape == (1, 16, 1)
    with pytest.raises(AssertionError):
        R.init_latents(1, 128, 8, "rel_pos_periodic")       # LU:88: 128 is not a square


def test_unknown_invariant_raises():
    with pytest.raises(ValueError):
        R.invariant_spec("nope")
    with pytest.raises(AssertionError):
        R.invariant_spec("ponita", num_in=3)


def test_inner_loop_oracle_reduces_loss_and_matches_golden():
    g = np.load(os.path.join(GOLD, "inner_loop_ponita.npz"))
    cfg = make_cfg(invariant="ponita", D=64, H=2, C=16, O=1, freq=(0.05, 0.01))
    prm = T.to_torch(R.init_params(int(g["param_seed"]), cfg, jitter=float(g["jitter"])), torch.float64)
    t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    lat0 = {k[5:]: t64(g[k]) for k in g.files if k.startswith("lat0/")}
    lrs = {k[3:]: t64(g[k]) for k in g.files if k.startswith("lr/")}
    loss, fit = T.inner_loop(prm, cfg, lat0, lrs, t64(g["coords"]), t64(g["img"]), torch.tensor(g["masks"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-10
    for k, v in fit.items():
        assert np.abs(v.detach().numpy() - g["fit/" + k]).max() < 1e-9
    assert np.all(fit["gaussian_window"].detach().numpy() == lat0["gaussian_window"].numpy())   # TR:210-212
    # loss before any step, on the same final mask
    xs = t64(g["coords"])[g["masks"][:, -1]][None].expand(2, -1, -1)
    lat_b = {k: v.repeat_interleave(2, 0) for k, v in lat0.items()}
    out0 = T.nef_apply(prm, cfg, xs, T.split_pose(lat_b, R.invariant_spec("ponita")), lat_b["a"], lat_b["gaussian_window"])
    loss0 = ((out0 - t64(g["img"])[:, g["masks"][:, -1]]) ** 2).mean().item()
    assert loss.item() < loss0


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    kw, B, N, Z, seed, store_w = CASES[name]
    cfg = make_cfg(**kw)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    from tests.golden.make_golden import flatten, unflatten
    prm = R.init_params(int(g["param_seed"]), cfg, jitter=float(g["jitter"]))
    stored = unflatten(g)
    if stored is not None and name != "tiny_rel_pos_periodic":
        # the fixture carries its weights: outputs and gradients belong to exactly those (fp32-representable) values,
        # independent of init_params' random stream
        out = R.nef_apply(stored, cfg, g["x"], g["p"], g["a"], g["sigma"])
        assert np.abs(out - g["out"]).max() < 1e-12
        tp = T.to_torch(stored, torch.float64)
        tpp, ta, ts = (torch.tensor(g[k], requires_grad=True) for k in ("p", "a", "sigma"))
        (T.nef_apply(tp, cfg, torch.tensor(g["x"]), tpp, ta, ts) * torch.tensor(g["w"])).sum().backward()
        for key, ten in (("dp", tpp), ("da", ta), ("dsigma", ts)):
            assert np.abs(ten.grad.numpy() - g[key]).max() < 1e-9 * max(1.0, np.abs(g[key]).max()), key
    else:
        out = R.nef_apply(prm, cfg, g["x"], g["p"], g["a"], g["sigma"])
        assert np.abs(out - g["out"]).max() < 1e-12
    if store_w:   # the stored fp32 weights are the seeded weights
        assert stored is not None
        for k, v in flatten(prm["params"]).items():
            assert np.allclose(g["W/" + k], v, rtol=1e-6, atol=1e-7)


def test_golden_files_are_all_covered():
    from tests.golden.make_golden import ODE_CASES
    files = {os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "*.npz"))}
    assert files == set(CASES) | set(ODE_CASES) | {"inner_loop_ponita", "config1_trace"}


def test_config1_full_size_trace_is_reproduced_by_the_numpy_oracle():
    """tests/golden/config1_trace.npz (BASELINE.json config 1 at full size, written by the torch restatement): the
    INDEPENDENT numpy restatement decodes the fitted latents to the same field, the stored loss is the loss of that field on
    the last mask, and the inner loop moved the latents (a, poses) but not the frozen window."""
    g = np.load(os.path.join(GOLD, "config1_trace.npz"))
    cfg = make_cfg(invariant="ponita", D=64, H=2, C=16, O=1, freq=(0.05, 0.01))
    prm = R.init_params(int(g["param_seed"]), cfg, jitter=float(g["jitter"]))
    B, N = g["img"].shape[:2]
    assert (B, N, g["fit/a"].shape[1], g["masks"].shape) == (8, 32 * 32, 16, (1024, 4))
    pose = np.concatenate([g["fit/p_pos"], g["fit/p_ori"]], -1)
    half = slice(0, 4)                                         # half the batch keeps the (B, N, Z, .) intermediates small
    out = R.nef_apply(prm, cfg, np.repeat(g["coords"][None], 4, 0), pose[half], g["fit/a"][half], g["fit/gaussian_window"][half])
    assert np.abs(out - g["recon"][half]).max() < 1e-9
    last = g["masks"][:, -1]
    assert abs(((g["recon"][:, last] - g["img"][:, last]) ** 2).mean() - float(g["loss"])) < 1e-12
    assert np.abs(g["fit/a"] - g["lat0/a"]).max() > 1e-3 and np.abs(g["fit/p_pos"] - g["lat0/p_pos"]).max() > 1e-5
    assert np.all(g["fit/gaussian_window"] == g["lat0/gaussian_window"])
