"""Backward-to-latents parity: HIP d/d(p, a, sigma) through the C-ABI against fp64 autograd of the
torch oracle (the reference obtains these from jax.grad, pde_trainer.py:188,200)."""
import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from oracle import enf_ref_torch as T
from tests.helpers import make_cfg, make_inputs, build_nef

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("bwd_variant")]

# relative L2 error of each gradient tensor
TOL = {"f32": 2e-4, "bf16": 7e-2}


def ref_grads(prm, cfg, x, p, a, s, w):
    tp = T.to_torch(prm, torch.float64)
    tx = torch.tensor(x)
    tpp, ta, ts = (torch.tensor(v, requires_grad=True) for v in (p, a, s))
    out = T.nef_apply(tp, cfg, tx, tpp, ta, ts)
    (out * torch.tensor(w)).sum().backward()
    z = lambda t: np.zeros(t.shape) if t.grad is None else t.grad.numpy()
    return out.detach().numpy(), z(tpp), z(ta), z(ts)


def hip_grads(cuda, nef, prm, x, p, a, s, w):
    params = nef.load_params(prm, device=cuda)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    tp, ta, ts = t(p, True), t(a, True), t(s, True)
    out = nef.apply(params, t(x), tp, ta, ts)
    (out * t(w)).sum().backward()
    torch.cuda.synchronize()
    g = lambda v: np.zeros(tuple(v.shape)) if v.grad is None else v.grad.cpu().numpy().astype(np.float64)
    return out.detach().cpu().numpy(), g(tp), g(ta), g(ts)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def check(cuda, cfg, B, N, Z, precision, seed=0, check_sigma=True):
    prm = R.init_params(seed, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, B, N, Z, seed + 1)
    w = np.random.default_rng(seed + 2).standard_normal((B, N, cfg["num_out"]))
    _, rp, ra, rs = ref_grads(prm, cfg, x, p, a, s, w)
    nef = build_nef(cfg, precision)
    _, gp, ga, gs = hip_grads(cuda, nef, prm, x, p, a, s, w)
    # a reference gradient that vanishes identically (one latent: the softmax no longer sees its logit, so d/d sigma and the
    # logit path of d/dp are exactly 0) is compared on the scale of the feature gradient instead of its own
    scale = np.linalg.norm(ra)
    relz = lambda g, r: rel(g, r) if np.linalg.norm(r) > 1e-9 * scale else np.linalg.norm(g) / scale
    errs = {"p": relz(gp, rp), "a": rel(ga, ra)}
    if check_sigma and cfg.get("use_gaussian_window", True):
        errs["sigma"] = relz(gs, rs)
    for k, e in errs.items():
        assert np.isfinite(e) and e < TOL[precision], (cfg["invariant"], precision, k, e, errs)
    return errs


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("invariant", ["rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos",
                                       "rel_pos", "norm_rel_pos"])
def test_backward_invariants(cuda, invariant, precision):
    cfg = make_cfg(invariant, D=128, H=2, C=16, O=3, freq=(0.5, 1.0))
    check(cuda, cfg, B=2, N=70, Z=9, precision=precision)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("D,H,C,O,Z,N", [(128, 2, 16, 1, 64, 512), (64, 2, 16, 1, 16, 100), (128, 1, 32, 3, 18, 33),
                                         (64, 1, 8, 2, 4, 32), (128, 2, 16, 1, 3, 40)])
def test_backward_shapes(cuda, D, H, C, O, Z, N, precision):
    cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=C, O=O)
    check(cuda, cfg, B=3, N=N, Z=Z, precision=precision, seed=D + Z)


_FULL_REF = {}


def _full_size_case(invariant, C, O, Z, N, seed):
    """oracle gradients of one full-size case, computed once per session (fp64 autograd over N * Z pairs: seconds)"""
    key = (invariant, C, O, Z, N, seed)
    if key not in _FULL_REF:
        cfg = make_cfg(invariant, D=128, H=2, C=C, O=O, freq=(0.05, 0.2))
        prm = R.init_params(seed, cfg, jitter=0.1)
        x, p, a, s = make_inputs(cfg, 1, N, Z, seed + 1)
        w = np.random.default_rng(seed + 2).standard_normal((1, N, O))
        _FULL_REF[key] = (cfg, prm, (x, p, a, s, w), ref_grads(prm, cfg, x, p, a, s, w))
    return _FULL_REF[key]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("invariant,C,O", [("latitude_periodic", 32, 3), ("polar_periodic", 32, 3), ("rel_pos_periodic", 16, 1)])
def test_backward_at_bench_latent_count(cuda, invariant, C, O, precision):
    """The backward the bench runs for BASELINE configs 3 and 4, at their own shape: 128 latents, num_hidden 128, two heads,
    512 sampled points (config_shallow_water.yaml:39-55,73: latitude_periodic, latent_dim 32, 3 fields; polar_periodic is the
    SO(3) bi-invariant of the same shape; config 4 = rel_pos_periodic with 128 latents), against fp64 autograd of the oracle
    under both backward kernel variants -- the z-fold kernel with its per-latent panels at Z = 128 included."""
    cfg, prm, (x, p, a, s, w), (ro, rp, ra, rs) = _full_size_case(invariant, C, O, 128, 512, 77)
    nef = build_nef(cfg, precision)
    out, gp, ga, gs = hip_grads(cuda, nef, prm, x, p, a, s, w)
    assert np.abs(out - ro).max() / np.abs(ro).max() < (2e-5 if precision == "f32" else 3e-2)
    for k, g, r in (("p", gp, rp), ("a", ga, ra), ("sigma", gs, rs)):
        assert rel(g, r) < TOL[precision], (invariant, precision, k, rel(g, r))


def test_backward_no_window(cuda):
    cfg = make_cfg("rel_pos", use_window=False, freq=(0.5, 0.5))
    check(cuda, cfg, B=2, N=64, Z=8, precision="f32")


@pytest.mark.parametrize("case", range(10))
def test_random_shape_sweep(cuda, case, pair_variant, bwd_variant):
    """Seeded random shapes (ragged N and Z, every invariant, widths / heads / precisions mixed): outputs and latent gradients
    against the oracle under all four combinations of forward / backward kernel variants."""
    rng = np.random.default_rng(1000 + case)
    inv = ["rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos", "rel_pos", "norm_rel_pos", "ball",
           "ball_lat"][case % 9]
    D, H = [(64, 1), (64, 2), (128, 1), (128, 2), (64, 4)][int(rng.integers(5))]
    if inv in ("ball", "ball_lat"):
        D = 64
        H = min(H, 2) if D == 64 and H == 4 else H
    B, N, Z = int(rng.integers(1, 4)), int(rng.integers(1, 150)), int(rng.integers(1, 40))
    precision = "f32" if case % 2 == 0 else "bf16"
    cfg = make_cfg(inv, D=D, H=H, C=int(rng.integers(2, 20)), O=int(rng.integers(1, 5)), freq=(0.3, 0.6))
    check(cuda, cfg, B=B, N=N, Z=Z, precision=precision, seed=2000 + case)


@pytest.mark.parametrize("H", [1, 2, 4])
def test_backward_is_reproducible(cuda, H, bwd_variant):
    """The same inputs give the same gradients run after run (up to the order of the atomic adds), with other work --
    and other contents of freed memory -- in between: the unfolded 64-wide bf16 kernel with two heads once differed by 1e-2
    between runs (a start-up race, scripts/k3_race/diag_k3_unfolded.py), far inside the oracle tolerance of this file."""
    cfg = make_cfg("ponita", D=64, H=H, C=7, O=2, freq=(0.3, 0.6))
    prm = R.init_params(5, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 1, 87, 11, 6)
    w = np.random.default_rng(7).standard_normal((1, 87, cfg["num_out"]))
    first = None
    for it in range(12):
        junk = [torch.randn(int(n), device=cuda) * 10 for n in np.random.default_rng(it).integers(1 << 10, 1 << 21, 8)]
        del junk
        res = hip_grads(cuda, build_nef(cfg, "bf16"), prm, x, p, a, s, w)
        if first is None:
            first = res
            continue
        for name, r0, r1 in zip(("out", "dp", "da", "dsigma"), first, res):
            assert rel(r1, r0) < 1e-5, (name, it, rel(r1, r0))


@pytest.mark.parametrize("D,H,precision", [(64, 2, "bf16"), (64, 1, "bf16"), (128, 2, "bf16"), (128, 1, "bf16"), (64, 2, "f32")])
def test_duplicate_waves_agree(cuda, D, H, precision):
    """In-launch determinism of the unfolded backward pair kernel.  With B*Z = 2 latents the waves 2..7 of every workgroup are
    inactive and recompute latent 1 to keep the barrier cadence, so inside ONE launch six waves must reproduce wave 1's final
    per-latent sums bit for bit.  Needs the test library (libenf_hip_test.so: the kernel's epilogue dumps every wave's sums;
    the tile loop is the product's).  This check caught what the run-to-run comparison above only saw on some GPUs: VALU reads
    of MFMA results without the wait states gfx950 needs ("K3 run-to-run deviations")."""
    import ctypes
    from enf_pde_amd import _lib
    tl = _lib.load_test()
    cfg = make_cfg("ponita", D=D, H=H, C=7, O=2, freq=(0.3, 0.6))
    prm = R.init_params(5, cfg, jitter=0.1)
    N = 1400
    x, p, a, s = make_inputs(cfg, 1, N, 2, 6)
    w = np.random.default_rng(7).standard_normal((1, N, cfg["num_out"]))
    row = 16 * 64 + 16
    nef = build_nef(cfg, precision)
    nef.pair_variants = ("latent_split", "latent_split")
    with _lib.using(tl):
        for it in range(4):
            hip_grads(cuda, nef, prm, x, p, a, s, w)
            buf = (ctypes.c_float * (64 * 8 * row))()
            assert tl.enf_test_read_wave_sums(buf) == 0
            sums = np.array(buf, dtype=np.float32).reshape(64, 8, row)
            launched = sums[:, 0, 16 * 64 + 10] > 0                       # tiles swept by wave 0 of the workgroup
            assert launched.sum() >= 32
            assert (sums[launched][:, 1:, 16 * 64 + 9] == 1).all()         # every wave from 1 on worked on latent 1
            ref = sums[launched][:, 1:2, :16 * 64 + H + 5]
            dup = sums[launched][:, 2:, :16 * 64 + H + 5]
            differ = (dup != ref).any(-1)
            assert not differ.any(), (it, int(differ.sum()), "duplicate waves differ from wave 1; per wave", differ.sum(0).tolist())


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("invariant,N,Z,O", [("rel_pos_periodic", 300, 20, 1), ("ponita", 77, 5, 2), ("latitude_periodic", 130, 9, 3)])
def test_fit_step_in_one_call_matches_the_three_calls_and_autograd(cuda, invariant, N, Z, O, precision, monkeypatch):
    """enf_fit_step (one inner step: forward, mean squared error and its gradient inside the tail kernel, backward to the latents)
    against the same step as enf_forward_stages + enf_mse_value_grad + enf_backward_latents_ex, and against autograd of
    mean((nef.apply - target)^2) through the model API (pde_trainer.py:175-207)."""
    from enf_pde_amd.enf import models as M
    cfg = make_cfg(invariant, D=128 if invariant != "ponita" else 64, H=2, C=12, O=O, freq=(0.3, 0.6))
    prm = R.init_params(17, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 3, N, Z, 18)
    y = np.random.default_rng(19).standard_normal((3, N, O))
    nef = build_nef(cfg, precision)
    params = nef.load_params(prm, device=cuda)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(M, "FUSED_FIT_STEP", fused)
        res[fused] = nef.mse_value_and_latent_grads(params, t(x), t(p), t(a), t(s), t(y), grad_scale=3.0)
    tp, ta, ts = t(p, True), t(a, True), t(s, True)
    loss = ((nef.apply(params, t(x), tp, ta, ts) - t(y)) ** 2).mean()
    (3.0 * loss).backward()
    tol = 1e-5 if precision == "f32" else 2e-2
    for other in (res[False], (loss.detach().reshape(1), tp.grad, ta.grad, ts.grad)):
        assert abs(float(res[True][0]) - float(other[0])) < tol * float(other[0])
        for g, r in zip(res[True][1:], other[1:]):
            assert rel(g.cpu().numpy(), r.cpu().numpy()) < 10 * tol
