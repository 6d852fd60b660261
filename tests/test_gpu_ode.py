"""Latent ODE on the GPU: the fused separable group convolution (csrc/enf_ode.hip) against a plain
PyTorch reference of the same op, and PonitaODEGen / MLPODE / the solvers against the oracle (fp64), values and
gradients w.r.t. latents and weights."""
import numpy as np
import pytest
import torch
from types import SimpleNamespace as NS

from oracle import enf_ref_torch as T
from oracle import ode_ref_np as O
from oracle import ode_ref_torch as OT
from tests.test_ode_oracle import ode_cfg, ode_inputs

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("B,Z,J,C,bias", [(3, 7, 16, 16, True), (2, 16, 64, 128, True), (2, 64, 64, 128, False),
                                          (1, 25, 128, 32, True), (4, 9, 32, 64, True), (2, 33, 128, 128, True)])
def test_sep_gconv_matches_einsum(cuda, B, Z, J, C, bias):
    from enf_pde_amd.fitting.ode_models import sep_gconv
    g = torch.Generator().manual_seed(Z + J)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    a, kb, W, b_, w = mk(B, Z, C), mk(B, Z, Z, J), mk(J, C) / J ** 0.5, mk(C), mk(B, Z, C)
    ref_in = [t.clone().requires_grad_(True) for t in (a, kb, W, b_)]
    ref = torch.einsum("bsc,brsc->brc", ref_in[0], ref_in[1] @ ref_in[2]) + (ref_in[3] if bias else 0)   # ponita_ode_g.py:72-82
    (ref * w).sum().backward()
    dev_in = [t.to(cuda, torch.float32).requires_grad_(True) for t in (a, kb, W, b_)]
    out = sep_gconv(dev_in[0], dev_in[1], dev_in[2], dev_in[3] if bias else None)
    (out * w.to(cuda, torch.float32)).sum().backward()
    torch.cuda.synchronize()
    assert rel(out.detach().cpu().double().numpy(), ref.detach().numpy()) < 1e-5
    for i, name in enumerate(["a", "kb", "W", "bias"][:4 if bias else 3]):
        assert rel(dev_in[i].grad.cpu().double().numpy(), ref_in[i].grad.numpy()) < 2e-5, name


@pytest.mark.parametrize("I,degree", [(4, 3), (3, 3), (5, 3), (1, 3), (2, 2), (6, 1)])
def test_poly_features_match_definition(cuda, I, degree):
    """enf_ode_poly_forward / _backward against the reference's definition (einsum chain, ponita_ode_g.py:22-26) in fp64."""
    from enf_pde_amd.fitting.ode_models import PolynomialFeatures
    g = torch.Generator().manual_seed(I * 10 + degree)
    x = torch.randn(3, 5, 7, I, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    ref = OT.poly_features(xr, degree)
    w = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * w).sum().backward()
    xd = x.to(cuda, torch.float32).requires_grad_(True)
    out = PolynomialFeatures(degree)(xd)
    assert out.shape == ref.shape == (3, 5, 7, O.num_poly_features(I, degree))
    (out * w.to(cuda, torch.float32)).sum().backward()
    assert rel(out.detach().cpu().double().numpy(), ref.detach().numpy()) < 1e-6
    assert rel(xd.grad.cpu().double().numpy(), xr.grad.numpy()) < 1e-5


@pytest.mark.parametrize("I,H1,J,P", [(4, 128, 64, 16 * 64 * 64 // 8), (4, 128, 64, 1000), (3, 64, 64, 777), (1, 32, 32, 130),
                                      (2, 64, 128, 257), (4, 32, 64, 64), (3, 128, 128, 1500), (4, 64, 32, 5)])
def test_kernel_basis_fused_matches_definition(cuda, I, H1, J, P):
    """enf_ode_basis_forward / _backward (csrc/enf_ode_basis.hip) against the definition in fp64: PolynomialFeatures
    (ponita_ode_g.py:15-26) -> Dense -> gelu -> Dense -> gelu (:128-131, 158-160); values, d inv and the four weight gradients;
    pair counts that are not multiples of the kernels' tiles."""
    from enf_pde_amd.fitting.ode_models.ponita_ode_g import kernel_basis
    g = torch.Generator().manual_seed(I * 1000 + H1 + J + P)
    F = O.num_poly_features(I, 3)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x, W1, b1, W3, b3, w = mk(P, I) * 0.7, mk(F, H1) / F ** 0.5, mk(H1) * 0.3, mk(H1, J) / H1 ** 0.5, mk(J) * 0.3, mk(P, J)
    ref_in = [t.clone().requires_grad_(True) for t in (x, W1, b1, W3, b3)]
    gelu = lambda t: torch.nn.functional.gelu(t, approximate="tanh")
    ref = gelu(gelu(OT.poly_features(ref_in[0], 3) @ ref_in[1] + ref_in[2]) @ ref_in[3] + ref_in[4])
    (ref * w).sum().backward()
    dev_in = [t.to(cuda, torch.float32).requires_grad_(True) for t in (x, W1, b1, W3, b3)]
    out = kernel_basis(dev_in[0], 3, {"kernel": dev_in[1], "bias": dev_in[2]}, {"kernel": dev_in[3], "bias": dev_in[4]})
    assert type(out.grad_fn).__name__ == "_KernelBasisBackward"                     # the fused path, not the fallback
    (out * w.to(cuda, torch.float32)).sum().backward()
    torch.cuda.synchronize()
    assert rel(out.detach().cpu().double().numpy(), ref.detach().numpy()) < 2e-6
    for i, name in enumerate(["inv", "W1", "b1", "W3", "b3"]):
        assert rel(dev_in[i].grad.cpu().double().numpy(), ref_in[i].grad.numpy()) < 1e-5, name
    # bitwise reproducible (fixed-order reduction of the workgroup partials)
    again = [t.detach().clone().requires_grad_(True) for t in dev_in]
    o2 = kernel_basis(again[0], 3, {"kernel": again[1], "bias": again[2]}, {"kernel": again[3], "bias": again[4]})
    (o2 * w.to(cuda, torch.float32)).sum().backward()
    assert torch.equal(o2, out) and all(torch.equal(a.grad, b.grad) for a, b in zip(again, dev_in))


def test_kernel_basis_wide_hidden_and_fallbacks(cuda):
    """hidden 256 (config_shallow_water.yaml's node) is fused for inference and takes the unfused path for training;
    I > 4 (ball) always does; both agree with the fused / unfused path of a supported shape."""
    from enf_pde_amd.fitting.ode_models import ponita_ode_g as M
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=g).to(cuda)
    for I, H1, J in [(4, 256, 128), (5, 64, 64), (4, 128, 64)]:
        F = O.num_poly_features(I, 3)
        x, K1, K3 = mk(300, I) * 0.7, {"kernel": mk(F, H1) / F ** 0.5, "bias": mk(H1) * 0.1}, {"kernel": mk(H1, J) / H1 ** 0.5, "bias": mk(J) * 0.1}
        with torch.no_grad():
            fused = M.kernel_basis(x, 3, K1, K3)
            M.FUSED_BASIS = False
            try:
                plain = M.kernel_basis(x, 3, K1, K3)
            finally:
                M.FUSED_BASIS = True
        assert rel(fused.cpu().double().numpy(), plain.cpu().double().numpy()) < 2e-6
        xg = x.clone().requires_grad_(True)
        out = M.kernel_basis(xg, 3, K1, K3)
        assert (type(out.grad_fn).__name__ == "_KernelBasisBackward") == (I <= 4 and H1 <= 128)
        out.sum().backward()
        assert torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("B,Z,H", [(16, 64, 128), (3, 7, 64), (2, 9, 32), (1, 33, 128), (5, 1000, 32)])
def test_latent_block_fused_matches_definition(cuda, B, Z, H):
    """enf_ode_block_forward / _backward (csrc/enf_ode_block.hip) against the definition in fp64 -- LayerNorm(1e-6) -> Dense ->
    gelu -> Dense (ConvBlock, ponita_ode_g.py:44-48): values and every gradient; row counts that are not multiples of 16; the
    library-GEMM path of the same autograd node (other widths) agrees too; two runs bitwise equal.  (5, 1000, 32): 5000 rows,
    more than one backward chunk of 4096 rows -- the weight gradients accumulate over chunks in a scratch of bounded size."""
    from enf_pde_amd.fitting.ode_models import ponita_ode_g as M_
    g = torch.Generator().manual_seed(B * 100 + Z + H)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    M = 2 * H
    x, gamma, beta, W1, b1, W2, b2, w = (mk(B, Z, H) * 2 + 0.5, 1 + 0.2 * mk(H), 0.2 * mk(H), mk(H, M) / H ** 0.5, 0.2 * mk(M),
                                       mk(M, H) / M ** 0.5, 0.2 * mk(H), mk(B, Z, H))
    ref_in = [t.clone().requires_grad_(True) for t in (x, gamma, beta, W1, b1, W2, b2)]
    xn = torch.nn.functional.layer_norm(ref_in[0], (H,), ref_in[1], ref_in[2], 1e-6)
    ref = torch.nn.functional.gelu(xn @ ref_in[3] + ref_in[4], approximate="tanh") @ ref_in[5] + ref_in[6]
    (ref * w).sum().backward()
    outs = []
    for fused in (True, False):
        M_.FUSED_BLOCK = fused
        try:
            dev_in = [t.to(cuda, torch.float32).requires_grad_(True) for t in (x, gamma, beta, W1, b1, W2, b2)]
            out = M_._LatentMLP.apply(*dev_in)
            (out * w.to(cuda, torch.float32)).sum().backward()
        finally:
            M_.FUSED_BLOCK = True
        torch.cuda.synchronize()
        assert rel(out.detach().cpu().double().numpy(), ref.detach().numpy()) < 2e-6
        for i, name in enumerate(["x", "gamma", "beta", "W1", "b1", "W2", "b2"]):
            assert rel(dev_in[i].grad.cpu().double().numpy(), ref_in[i].grad.numpy()) < 1e-5, (fused, name)
        outs.append((out, [t.grad for t in dev_in]))
    again = [t.to(cuda, torch.float32).requires_grad_(True) for t in (x, gamma, beta, W1, b1, W2, b2)]
    o2 = M_._LatentMLP.apply(*again)
    (o2 * w.to(cuda, torch.float32)).sum().backward()
    assert torch.equal(o2, outs[0][0]) and all(torch.equal(a.grad, b) for a, b in zip(again, outs[0][1]))


@pytest.mark.parametrize("B,Z,I,D,cr,cs", [(16, 64, 4, 2, 1.0, -1.0), (3, 7, 3, 2, 0.0, 1.0), (2, 70, 5, 3, 1.0, -1.0), (1, 130, 1, 2, 1.0, -1.0)])
def test_vec_readout_fused_matches_definition(cuda, B, Z, I, D, cr, cs):
    """enf_ode_vec_readout_forward / _backward against the op-by-op definition (ponita_ode_g.py:176-193) in fp64: values and the
    gradients w.r.t. invariants, per-latent weights, receiver / sender vectors and the invariant rows of the readout kernel."""
    from enf_pde_amd.fitting.ode_models.ponita_ode_g import _VecReadout
    g = torch.Generator().manual_seed(B + Z + I)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    inv, aw, u, w_, Wi, ct = mk(B, Z, Z, I), mk(B, Z), mk(B, Z, D), mk(B, Z, D), mk(I), mk(B, Z, D)
    ref_in = [t.clone().requires_grad_(True) for t in (inv, aw, u, w_, Wi)]
    wgt = (ref_in[0] * ref_in[4]).sum(-1) + ref_in[1][:, None, :]
    ref = (wgt[..., None] * (cr * ref_in[2][:, :, None, :] + cs * ref_in[3][:, None, :, :])).mean(-2)
    (ref * ct).sum().backward()
    dev_in = [t.to(cuda, torch.float32).requires_grad_(True) for t in (inv, aw, u, w_, Wi)]
    out = _VecReadout.apply(*dev_in, cr, cs)
    (out * ct.to(cuda, torch.float32)).sum().backward()
    torch.cuda.synchronize()
    assert rel(out.detach().cpu().double().numpy(), ref.detach().numpy()) < 2e-6
    for i, name in enumerate(["inv", "aw", "u", "w", "Wi"]):
        if name == "u" and cr == 0.0:
            assert float(dev_in[i].grad.abs().max()) == 0.0
            continue
        assert rel(dev_in[i].grad.cpu().double().numpy(), ref_in[i].grad.numpy()) < 1e-5, name


def test_sep_gconv_rejects_unsupported(cuda):
    from enf_pde_amd.fitting.ode_models import sep_gconv
    z = lambda *s: torch.zeros(*s, device=cuda)
    with pytest.raises(NotImplementedError):
        sep_gconv(z(1, 4, 24), z(1, 4, 4, 16), z(16, 24))          # C not in {16, 32, 64, 128}
    with pytest.raises(RuntimeError):
        sep_gconv(torch.zeros(1, 4, 16), torch.zeros(1, 4, 4, 16), torch.zeros(16, 16))    # no CPU path


def _flat(tree, prefix=""):
    for k in sorted(tree):
        if isinstance(tree[k], dict):
            yield from _flat(tree[k], prefix + k + "/")
        else:
            yield prefix + k, tree[k]


def _model(cfg, latent_dim):
    from enf_pde_amd.enf.steerable_attention.invariant import get_sa_invariant
    from enf_pde_amd.fitting.ode_models import PonitaODEGen
    inv = get_sa_invariant(NS(invariant_type=cfg["invariant"], num_in=cfg["num_in"]))
    return PonitaODEGen(num_hidden=cfg["num_hidden"], num_layers=cfg["num_layers"], scalar_num_out=latent_dim, vec_num_out=1,
                        invariant=inv, basis_dim=cfg["basis_dim"], degree=cfg["degree"], widening_factor=cfg["widening_factor"],
                        global_pool=False, kernel_size=cfg["kernel_size"])


@pytest.mark.parametrize("inv,Z,C,hid,basis,ks", [("rel_pos_periodic", 16, 16, 128, 64, "global"),      # config_navier_stokes.yaml
                                                  ("ponita", 9, 32, 128, 128, 0.2),                      # config_cahn_hilliard.yaml
                                                  ("polar_periodic", 18, 4, 32, 32, "global"),           # config_diff_sphere.yaml
                                                  ("latitude_periodic", 8, 32, 128, 64, "global"),       # config_shallow_water.yaml
                                                  ("ball", 25, 32, 128, 64, "global"),                   # config_ihc.yaml
                                                  ("rel_pos", 5, 8, 16, 16, "global")])
def test_ponita_ode_matches_oracle(cuda, inv, Z, C, hid, basis, ks):
    cfg = ode_cfg(inv, num_hidden=hid, basis_dim=basis, num_layers=3, kernel_size=ks)
    prm = O.init_ponita_ode(Z + C, cfg, latent_dim=C, jitter=0.1, readout_scale=1.0)
    lat = ode_inputs(cfg, 2, Z, C, Z)
    rng = np.random.default_rng(3)
    wp, wa = rng.standard_normal(lat[0].shape), rng.standard_normal(lat[1].shape)
    # oracle, fp64, autograd
    rp = T.to_torch(prm, torch.float64, requires_grad=True)
    rl = [torch.tensor(v, requires_grad=True) for v in lat[:2]] + [torch.tensor(lat[2])]
    odp, oda, odw = OT.ponita_ode(rp, cfg, tuple(rl))
    ((odp * torch.tensor(wp)).sum() + (oda * torch.tensor(wa)).sum()).backward()
    # product
    model = _model(cfg, C)
    P = model.load_params(prm, device=cuda)
    leaves = dict(_flat(P))
    for v in leaves.values():
        v.requires_grad_(True)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    p, a, w = t(lat[0], True), t(lat[1], True), t(lat[2])
    dp, da, dw = model.apply(P, (p, a, w))
    ((dp * t(wp)).sum() + (da * t(wa)).sum()).backward()
    torch.cuda.synchronize()
    n = lambda v: v.detach().cpu().double().numpy()
    assert dp.shape == p.shape and da.shape == a.shape and not n(dw).any()
    assert rel(n(dp), odp.detach().numpy()) < 2e-4 and rel(n(da), oda.detach().numpy()) < 2e-4
    assert rel(n(p.grad), rl[0].grad.numpy()) < 1e-3 and rel(n(a.grad), rl[1].grad.numpy()) < 1e-3
    ref_leaves = dict(_flat(rp))
    assert set(ref_leaves) == set(leaves)
    for k, v in leaves.items():
        assert rel(n(v.grad), ref_leaves[k].grad.numpy()) < 2e-3, k


def test_init_shapes_and_flax_names(cuda):
    cfg = ode_cfg("ponita", num_hidden=32, basis_dim=16, num_layers=3)
    model = _model(cfg, 8)
    lat = tuple(torch.tensor(v, dtype=torch.float32, device=cuda) for v in ode_inputs(cfg, 1, 4, 8, 0))
    P = model.init(0, lat)
    ref = O.init_ponita_ode(0, cfg, latent_dim=8)
    got, want = dict(_flat(P)), dict(_flat(ref))
    assert {k: tuple(v.shape) for k, v in got.items()} == {k: tuple(v.shape) for k, v in want.items()}
    k1 = got["params/ponita/kernel_basis/layers_1/kernel"]
    assert abs(float(k1.std()) - (1 / 120) ** 0.5) < 0.15 * (1 / 120) ** 0.5         # lecun_normal over 120 polynomial features
    assert float(got["params/ponita/readout_scalar/layers_0/kernel"].abs().max()) < 1e-2   # variance_scaling(1e-6)
    dp, da, dw = model.apply(P, lat)                                                  # near-zero derivative at init
    assert dp.shape == (1, 4, 3) and da.shape == (1, 4, 8) and float(da.abs().max()) < 1e-1


@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_rollout_matches_oracle(cuda, method):
    from enf_pde_amd.fitting.trainers.trainer_utils import solve_latent_ode
    cfg = ode_cfg("rel_pos_periodic", num_hidden=32, basis_dim=16, num_layers=2)
    prm = O.init_ponita_ode(5, cfg, latent_dim=8, jitter=0.1, readout_scale=0.05)
    lat = ode_inputs(cfg, 2, 6, 8, 6)
    ref = O.solve_latent_ode(lambda z, t: O.ponita_ode(prm, cfg, z), lat, 0, 4, 1, method=method)
    model = _model(cfg, 8)
    P = model.load_params(prm, device=cuda)
    dl = tuple(torch.tensor(v, dtype=torch.float32, device=cuda) for v in lat)
    with torch.no_grad():
        got = solve_latent_ode(lambda z, t: model.apply(P, z), dl, 0, 4, 1, method=method)
    for g, r in zip(got, ref):
        assert tuple(g.shape) == r.shape and r.shape[1] == 5
        assert rel(g.cpu().double().numpy(), r) < 2e-4
    assert np.abs(ref[1][:, -1] - ref[1][:, 0]).max() > 1e-3          # the latents do move


def test_graphed_derivative_replays_bitwise(cuda):
    """PonitaODEGen.graphed: one captured hipGraph per derivative evaluation gives the same roll-out as eager launches."""
    from enf_pde_amd.fitting.trainers.trainer_utils import solve_latent_ode
    cfg = ode_cfg("rel_pos_periodic", num_hidden=64, basis_dim=32, num_layers=3)
    prm = O.init_ponita_ode(9, cfg, latent_dim=8, jitter=0.1, readout_scale=0.05)
    model = _model(cfg, 8)
    P = model.load_params(prm, device=cuda)
    dl = tuple(torch.tensor(v, dtype=torch.float32, device=cuda) for v in ode_inputs(cfg, 3, 16, 8, 10))
    with torch.no_grad():
        f = model.graphed(P, dl)
        eager = solve_latent_ode(lambda z, t: model.apply(P, z), dl, 0, 6, 1, method="rk4")
        graph = solve_latent_ode(lambda z, t: f(z), dl, 0, 6, 1, method="rk4")
    assert all(torch.equal(g, e) for g, e in zip(graph, eager))
    P["params"]["ponita"]["readout_scalar"]["layers_0"]["kernel"].mul_(2.0)        # parameters are read in place
    with torch.no_grad():
        assert torch.equal(f(dl)[1], model.apply(P, dl)[1])


def test_rollout_gradient_reaches_the_ode_weights(cuda):
    """ode_loss (pde_trainer.py:411-500) differentiates a roll-out w.r.t. the ODE parameters."""
    from enf_pde_amd.fitting.trainers.trainer_utils import solve_latent_ode
    cfg = ode_cfg("ponita", num_hidden=16, basis_dim=16, num_layers=1)
    prm = O.init_ponita_ode(7, cfg, latent_dim=4, readout_scale=0.05)
    lat = ode_inputs(cfg, 1, 4, 4, 8)
    rp = T.to_torch(prm, torch.float64, requires_grad=True)
    rt = OT.solve_latent_ode(lambda z, t: OT.ponita_ode(rp, cfg, z), tuple(torch.tensor(v) for v in lat), 0, 3, 1, method="euler")
    (rt[0] ** 2).sum().add((rt[1] ** 2).sum()).backward()
    model = _model(cfg, 4)
    P = model.load_params(prm, device=cuda)
    leaves = dict(_flat(P))
    for v in leaves.values():
        v.requires_grad_(True)
    dl = tuple(torch.tensor(v, dtype=torch.float32, device=cuda) for v in lat)
    gt = solve_latent_ode(lambda z, t: model.apply(P, z), dl, 0, 3, 1, method="euler")
    (gt[0] ** 2).sum().add((gt[1] ** 2).sum()).backward()
    ref_leaves = dict(_flat(rp))
    for k, v in leaves.items():
        assert rel(v.grad.cpu().double().numpy(), ref_leaves[k].grad.numpy()) < 2e-3, k


def test_mlp_ode_matches_oracle(cuda):
    from enf_pde_amd.fitting.ode_models import MLPODE
    prm = O.init_mlp_ode(13, 32, 2, 8)
    rng = np.random.default_rng(14)
    lat = (rng.uniform(-1, 1, (2, 6, 2)), 1 + 0.2 * rng.standard_normal((2, 6, 8)), np.ones((2, 6, 1)))
    rp, ra, _ = O.mlp_ode(prm, lat)
    m = MLPODE(num_hidden=32, num_layers=3, scalar_num_out=8, vec_num_out=1)
    dl = tuple(torch.tensor(v, dtype=torch.float32, device=cuda) for v in lat)
    dp, da, dw = m.apply(m.load_params(prm, device=cuda), dl)
    assert rel(dp.cpu().double().numpy(), rp) < 1e-5 and rel(da.cpu().double().numpy(), ra) < 1e-5 and not dw.any()
    shapes = {k: tuple(v.shape) for k, v in _flat(m.init(0, dl))}
    assert shapes == {k: tuple(v.shape) for k, v in _flat(prm)}


@pytest.mark.parametrize("name", ["ode_rel_pos_periodic", "ode_ponita"])
def test_ode_against_golden_fixtures(cuda, name):
    """tests/golden/ode_*.npz (frozen oracle vectors): derivative, its gradients for a fixed cotangent, Euler / RK4 traces."""
    import os
    from tests.golden.make_golden import ODE_CASES
    from enf_pde_amd.fitting.trainers.trainer_utils import solve_latent_ode
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    kw, B, Z, C, seed = ODE_CASES[name]
    cfg = ode_cfg(kw["invariant"], **{k: v for k, v in kw.items() if k != "invariant"})
    prm = O.init_ponita_ode(int(g["param_seed"]), cfg, latent_dim=C, jitter=float(g["jitter"]), readout_scale=float(g["readout_scale"]))
    model = _model(cfg, C)
    P = model.load_params(prm, device=cuda)
    t = lambda v, gr=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=gr)
    p, a, w = t(g["p"], True), t(g["a"], True), t(g["window"])
    dp, da, _ = model.apply(P, (p, a, w))
    ((dp * t(g["wp"])).sum() + (da * t(g["wa"])).sum()).backward()
    n = lambda v: v.detach().cpu().double().numpy()
    assert rel(n(dp), g["dp"]) < 2e-4 and rel(n(da), g["da"]) < 2e-4
    assert rel(n(p.grad), g["gp"]) < 1e-3 and rel(n(a.grad), g["ga"]) < 1e-3
    with torch.no_grad():
        for method in ("euler", "rk4"):
            tr = solve_latent_ode(lambda z, _: model.apply(P, z), (p.detach(), a.detach(), w), 0, 4, 1, method=method)
            assert rel(n(tr[0]), g[method + "/p"]) < 2e-4 and rel(n(tr[1]), g[method + "/a"]) < 2e-4, method
