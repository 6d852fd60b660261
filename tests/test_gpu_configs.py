"""BASELINE.json configs 4 and 5 at full size, through properties that need no oracle run (the oracle materialises
(B, N, Z, .) tensors: 128 latents x 128^2 queries is out of its reach).
  config 4: 2-D Navier-Stokes 128x128, 128 latents (one GPU's share of the meta-batch).
  config 5: fit on 64x64 -> 256x256 super-resolution decode + 40-step latent roll-out."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from oracle import ode_ref_np as O
from tests.helpers import make_cfg, make_inputs, build_nef
from tests.test_ode_oracle import ode_cfg
from tests.test_gpu_ode import _model

pytestmark = pytest.mark.gpu


def _grid(n):
    lin = np.linspace(-1, 1, n)
    return np.stack(np.meshgrid(lin, lin), -1).reshape(-1, 2)


def test_config4_128_latents_128x128(cuda):
    from enf_pde_amd.fitting import inner_loop, decode, default_meta_sgd_lrs, make_masks
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(4, cfg, jitter=0.1)
    _, p, a, s = make_inputs(cfg, 2, 8, 128, 6)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    coords = t(_grid(128))
    nef, nb = build_nef(cfg, "f32"), build_nef(cfg, "bf16")
    P, Pb = nef.load_params(prm, device=cuda), nb.load_params(prm, device=cuda)
    base = decode(nef, P, coords, t(p), t(a), t(s))
    assert base.shape == (2, 128 * 128, 1) and torch.isfinite(base).all()
    # a random 2,000-point subset decoded on its own == the same points of the full decode (pointwise operator)
    idx = torch.randperm(128 * 128, generator=torch.Generator().manual_seed(0))[:2000].to(cuda)
    sub = nef.apply(P, coords[idx][None].expand(2, -1, -1), t(p), t(a), t(s))
    assert (sub - base[:, idx]).abs().max() < 2e-5 * base.abs().max()
    # the small-problem oracle agrees on those points for a 16-latent sub-problem (same weights)
    ref = R.nef_apply(prm, cfg, coords[idx][:200].cpu().numpy()[None].repeat(2, 0), p[:, :16], a[:, :16], s[:, :16])
    got = nef.apply(P, coords[idx][:200][None].expand(2, -1, -1), t(p[:, :16]), t(a[:, :16]), t(s[:, :16])).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()
    # latent permutation, joint translation
    perm = torch.randperm(128, generator=torch.Generator().manual_seed(1)).to(cuda)
    assert (decode(nef, P, coords, t(p)[:, perm], t(a)[:, perm], t(s)[:, perm]) - base).abs().max() < 3e-5 * base.abs().max()
    sh = torch.tensor([0.5, -0.25], device=cuda)
    assert (decode(nef, P, coords + sh, t(p) + sh, t(a), t(s)) - base).abs().max() < 5e-5 * base.abs().max()
    # bf16 field MSE budget (BASELINE.json: <= 1e-5)
    assert ((decode(nb, Pb, coords, t(p), t(a), t(s)) - base) ** 2).mean() < 1e-5
    # the inner loop at this size reduces the loss of a field the model can represent
    target = base.detach()
    lat0 = {"p_pos": t(p[:1]), "a": torch.ones(1, 128, 16, device=cuda), "gaussian_window": t(s[:1])}
    masks = make_masks(128 * 128, 1024, 3, generator=torch.Generator().manual_seed(2), device=cuda)
    l0 = ((decode(nef, P, coords, lat0["p_pos"], lat0["a"], lat0["gaussian_window"]) - target[:1]) ** 2).mean()
    _, lat = inner_loop(nef, P, lat0, default_meta_sgd_lrs(16, lr_p=0.0, lr_a=2.0, device=cuda), coords, target[:1], masks)
    l1 = ((decode(nef, P, coords, lat["p_pos"], lat["a"], lat["gaussian_window"]) - target[:1]) ** 2).mean()
    assert l1 < l0


def test_config5_superresolution_and_40_step_rollout(cuda):
    from enf_pde_amd.fitting import decode, solve_latent_ode
    cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
    prm = R.init_params(5, cfg, jitter=0.1)
    _, p, a, s = make_inputs(cfg, 2, 8, 64, 7)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    nef = build_nef(cfg, "bf16")
    P = nef.load_params(prm, device=cuda)
    # ---- 40-step latent roll-out (config_navier_stokes.yaml's node: ponita, hidden 128, basis 64, 3 layers, Euler dt 1)
    ocfg = ode_cfg("rel_pos_periodic", num_hidden=128, basis_dim=64, num_layers=3)
    oprm = O.init_ponita_ode(8, ocfg, latent_dim=16, jitter=0.05, readout_scale=1e-3)
    ode = _model(ocfg, 16)
    OP = ode.load_params(oprm, device=cuda)
    f = lambda z, _: ode.apply(OP, z)
    with torch.no_grad():
        traj = solve_latent_ode(f, (t(p), t(a), t(s)), 0, 40, 1, method="euler")
        assert traj[0].shape == (2, 41, 64, 2) and traj[1].shape == (2, 41, 64, 16) and all(torch.isfinite(v).all() for v in traj)
        assert torch.equal(traj[2][:, -1], t(s))                                  # no derivative for the window (:254-256)
        assert (traj[1][:, -1] - traj[1][:, 0]).abs().max() > 1e-4               # the latents move
        # composition: 40 steps == 20 steps, then 20 more from there (same launches -> bitwise)
        half = solve_latent_ode(f, (t(p), t(a), t(s)), 0, 20, 1, method="euler")
        rest = solve_latent_ode(f, tuple(v[:, -1] for v in half), 0, 20, 1, method="euler")
        assert all(torch.equal(r[:, -1], v[:, -1]) for r, v in zip(rest, traj))
        # equivariance of the roll-out: permuting the latents / translating every pose commutes with 40 steps
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(3)).to(cuda)
        tp = solve_latent_ode(f, (t(p)[:, perm], t(a)[:, perm], t(s)[:, perm]), 0, 40, 1, method="euler")
        assert (tp[1][:, -1] - traj[1][:, -1][:, perm]).abs().max() < 1e-4 and (tp[0][:, -1] - traj[0][:, -1][:, perm]).abs().max() < 1e-4
        sh = torch.tensor([0.3, -0.7], device=cuda)
        ts = solve_latent_ode(f, (t(p) + sh, t(a), t(s)), 0, 40, 1, method="euler")
        assert (ts[0][:, -1] - sh - traj[0][:, -1]).abs().max() < 1e-4 and (ts[1][:, -1] - traj[1][:, -1]).abs().max() < 1e-4
        # ---- 256 x 256 super-resolution decode of the rolled-out latents (last frame)
        pT, aT, sT = (v[:, -1].contiguous() for v in traj)
        c256, c64 = t(_grid(256)), t(_grid(64))
        hi = decode(nef, P, c256, pT, aT, sT)
        assert hi.shape == (2, 256 * 256, 1) and torch.isfinite(hi).all()
        # the operator is pointwise in the query: the 64x64 fit grid's corner points are also 256-grid points
        lo = decode(nef, P, c64, pT, aT, sT).view(2, 64, 64)
        hv = hi.view(2, 256, 256)
        for (i, j) in ((0, 0), (0, 63), (63, 0), (63, 63)):
            assert (hv[:, i * 255 // 63, j * 255 // 63] - lo[:, i, j]).abs().max() < 2e-2 * hi.abs().max()
        # any subset of the fine grid decoded alone gives the same values (bf16 mode: same arithmetic per query)
        idx = torch.randperm(256 * 256, generator=torch.Generator().manual_seed(4))[:4096].to(cuda)
        sub = nef.apply(P, c256[idx][None].expand(2, -1, -1), pT, aT, sT)
        assert (sub - hi[:, idx]).abs().max() < 2e-2 * hi.abs().max()
        # the decoded field is periodic like its invariant: shifting queries AND poses by the period 2 changes nothing
        assert (decode(nef, P, c256 + 2.0, pT + 2.0, aT, sT) - hi).abs().max() < 2e-2 * hi.abs().max()
