"""Latent-ODE phase of the trainer (pde_trainer.py:290-500): the roll-out loss and its gradient w.r.t. the ODE
parameters against the oracle (fp64 autograd through decoder + solver + ODE model), and the three step functions'
bookkeeping (which parameter groups move, optimiser rules, val_step's two errors)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from oracle import enf_ref_torch as T
from oracle import ode_ref_np as O
from oracle import ode_ref_torch as OT
from oracle import optim_ref_np as OP
from tests.helpers import make_cfg, build_nef
from tests.test_ode_oracle import ode_cfg
from tests.test_gpu_ode import _flat, _model, rel
from enf_pde_amd.fitting.trainers import MetaSGDPDETrainer
from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta

pytestmark = pytest.mark.gpu


def _setup(cuda, n_s=32, T_train=3, T_out=2, hidden=16, basis=16, method="euler"):
    cfg = make_cfg("rel_pos_periodic", D=64, H=2, C=8, O=1)
    prm = R.init_params(0, cfg, jitter=0.1)
    ocfg = ode_cfg("rel_pos_periodic", num_hidden=hidden, basis_dim=basis, num_layers=2)
    oprm = O.init_ponita_ode(1, ocfg, latent_dim=8, jitter=0.1, readout_scale=0.02)
    rng = np.random.default_rng(2)
    lin = np.linspace(-1, 1, 8)
    coords = np.stack(np.meshgrid(lin, lin), -1).reshape(-1, 2)
    traj = rng.standard_normal((2, T_train + T_out, 8, 8, 1))
    conf = NS(optimizer=NS(learning_rate_enf=1e-3, learning_rate_codes=0.0, learning_rate_ode=1e-3),
              meta=NS(learning_rate_meta_sgd=1e-2, num_inner_steps=2, inner_learning_rate_p=0.5, inner_learning_rate_a=2.0,
                      inner_learning_rate_window=0.0, noise_pos_inner_loop=0.0),
              nef=NS(optimize_gaussian_window=False), training=NS(max_num_sampled_points=n_s),
              node=NS(dt=1, method=method), dataset=NS(traj_len_train=T_train, traj_len_out_horizon=T_out))
    nef = build_nef(cfg, "f32")
    ode = _model(ocfg, 8)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    ad = PositionOrientationFeatureAutodecoderMeta(1, 9, 8, 2, 0, gaussian_window_size=-1)
    tr = MetaSGDPDETrainer(conf, nef, ad, t(coords), seed=0, second_order="fd", ode_model=ode)
    state = tr.init_train_state(nef.load_params(prm, device=cuda), ode_params=ode.load_params(oprm, device=cuda))
    return cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t


def test_ode_loss_and_gradient_match_oracle(cuda):
    cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t = _setup(cuda)
    rng = np.random.default_rng(5)
    lat = {"p_pos": R.init_positions_grid(2, 9, 2) + 0.05 * rng.standard_normal((2, 9, 2)),
           "a": 1 + 0.2 * rng.standard_normal((2, 9, 8)), "gaussian_window": np.full((2, 9, 1), 2.0 / 3)}
    pm = np.stack([rng.permutation(64)[:32] for _ in range(3)])
    # oracle: pde_trainer.py:429-481 in fp64
    rp = T.to_torch(oprm, torch.float64, requires_grad=True)
    tl = {k: torch.tensor(v, requires_grad=True) for k, v in lat.items()}
    sol = OT.solve_latent_ode(lambda z, _: OT.ponita_ode(rp, ocfg, z), (tl["p_pos"], tl["a"], tl["gaussian_window"]), 0, 2, 1, "euler")
    p_fl, a_fl, w_fl = (v.reshape(6, *v.shape[2:]) for v in sol)
    xs = torch.tensor(coords)[torch.tensor(pm)][None].expand(2, -1, -1, -1).reshape(6, 32, 2)
    tj = torch.tensor(traj[:, :3]).reshape(2, 3, 64, 1)
    ys = torch.stack([torch.stack([tj[b, k, pm[k]] for k in range(3)]) for b in range(2)]).reshape(6, 32, 1)
    ref = ((T.nef_apply(T.to_torch(prm, torch.float64), cfg, xs, p_fl, a_fl, w_fl) - ys) ** 2).mean()
    ref.backward()
    # product
    leaves = dict(_flat(state.params["ode_params"]))
    for v in leaves.values():
        v.requires_grad_(True)
    dl = {k: t(v).requires_grad_(True) for k, v in lat.items()}
    loss = tr.ode_loss(state.params["nef"], state.params["ode_params"], dl, t(traj[:, :3]), torch.tensor(pm, device=cuda))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-4 * float(ref.detach())
    ref_leaves = dict(_flat(rp))
    for k, v in leaves.items():
        assert rel(v.grad.cpu().double().numpy(), ref_leaves[k].grad.numpy()) < 5e-3, k
    for k in ("p_pos", "a"):
        assert rel(dl[k].grad.cpu().double().numpy(), tl[k].grad.numpy()) < 5e-3, k


@pytest.mark.parametrize("method,step", [("euler", "ode_train_step"), ("rk4", "ode_train_step"), ("euler", "dual_train_step")])
def test_captured_training_evaluations_equal_eager(cuda, method, step):
    """With ``training.graph_ode_training`` the ode / dual train steps replay one captured (forward, backward) hipGraph pair per
    derivative evaluation of the roll-out (PonitaODEGen.graphed_train; fused kernel basis inside: hidden 32, basis 32): three consecutive steps -- the optimiser
    hands out new parameter tensors, the graphs keep reading the trainer's persistent leaves -- give the losses and
    parameters of the eager path."""
    runs = []
    for graphs in (True, False):
        cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t = _setup(cuda, hidden=32, basis=32, method=method)
        tr.graph_ode_training = graphs
        batch = t(traj)
        losses = []
        for i in range(3):
            mk = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(10 + i))[:32] for _ in range(3)], 1).to(cuda)
            pm = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(20 + i))[:32] for _ in range(3)]).to(cuda)
            loss, state = getattr(tr, step)(state, batch, masks=mk, point_masks=pm)
            losses.append(float(loss))
        assert ("_ode_train_graphs" in tr.__dict__) == graphs
        runs.append((losses, [v.detach().clone() for _, v in _flat(state.params["ode_params"])]))
    (la, pa), (lb, pb) = runs
    # (the dual step also moves the nef weights, whose gradient sums atomically: run-to-run differences of a few 1e-6)
    tol = 1e-6 if step == "ode_train_step" else 5e-5
    assert np.allclose(la, lb, rtol=tol, atol=0), (la, lb)
    assert la[0] != la[2]                                                        # the parameters moved
    for a, b in zip(pa, pb):
        assert rel(a.cpu().double().numpy(), b.cpu().double().numpy()) < tol * 10


def test_ode_train_step_moves_only_the_ode(cuda):
    cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t = _setup(cuda)
    batch = t(traj)
    mk = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(1))[:32] for _ in range(3)], 1).to(cuda)
    pm = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(2))[:32] for _ in range(3)]).to(cuda)
    before = {k: v.clone() for k, v in _flat(state.params["ode_params"])}
    w0 = [w.clone() for w in tr.nef.param_tensors(state.params["nef"])]
    loss, new = tr.ode_train_step(state, batch, masks=mk, point_masks=pm)
    # Adam from a zero state: the first update is -lr * g / (|g| + eps')  (optax.adam; pde_trainer.py:66,304-306)
    lat = tr._fitted(state, batch[:, :3], mk)
    leaves = [v.detach().clone().requires_grad_(True) for _, v in _flat(state.params["ode_params"])]
    names = [k for k, _ in _flat(state.params["ode_params"])]
    from enf_pde_amd.fitting.trainers.pde_trainer import _unflatten
    l2 = tr.ode_loss(state.params["nef"], _unflatten(state.params["ode_params"], leaves), lat, batch[:, :3], pm)
    g = torch.autograd.grad(l2, leaves)
    assert abs(float(l2) - float(loss)) < 1e-6
    ref, _ = OP.adam_step([v.detach().cpu().numpy().astype(np.float64) for v in leaves], [x.cpu().numpy().astype(np.float64) for x in g],
                          OP.init_state([x.cpu().numpy() for x in g]), lr=1e-3)
    after = dict(_flat(new.params["ode_params"]))
    for k, r in zip(names, ref):
        np.testing.assert_allclose(after[k].detach().cpu().numpy(), r, rtol=2e-4, atol=2e-6)
        assert not torch.equal(after[k], before[k]) or float(before[k].abs().max()) == 0
    for a, b in zip(tr.nef.param_tensors(new.params["nef"]), w0):
        assert torch.equal(a, b)
    assert new.params["meta_sgd_lrs"] is state.params["meta_sgd_lrs"] and new.params["autodecoder"] is state.params["autodecoder"]
    losses = [float(loss)]
    st = new
    for _ in range(10):
        l, st = tr.ode_train_step(st, batch, masks=mk, point_masks=pm)
        losses.append(float(l))
    assert losses[-1] < losses[0], losses


def test_dual_train_step_and_val_step(cuda):
    cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t = _setup(cuda)
    batch = t(traj)
    mk = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(1))[:32] for _ in range(3)], 1).to(cuda)
    pm = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(2))[:32] for _ in range(3)]).to(cuda)
    w0 = [w.clone() for w in tr.nef.param_tensors(state.params["nef"])]
    o0 = {k: v.clone() for k, v in _flat(state.params["ode_params"])}
    l0 = {k: v.clone() for k, v in state.params["meta_sgd_lrs"].items()}
    loss, new = tr.dual_train_step(state, batch, masks=mk, point_masks=pm)
    assert np.isfinite(float(loss))
    assert any(not torch.equal(a, b) for a, b in zip(tr.nef.param_tensors(new.params["nef"]), w0))
    assert any(not torch.equal(v, o0[k]) for k, v in _flat(new.params["ode_params"]))
    assert not torch.equal(new.params["meta_sgd_lrs"]["a"], l0["a"])
    assert new.params["autodecoder"] is state.params["autodecoder"]               # pde_trainer.py:349 (not updated)
    # the same loss value as ode_train_step sees for the same state and masks
    loss_ode, _ = tr.ode_train_step(state, batch, masks=mk, point_masks=pm)
    assert abs(float(loss) - float(loss_ode)) < 1e-5 * max(1.0, abs(float(loss_ode)))
    # val_step against the oracle: fit frame 0, roll out 5 frames, decode the full grid
    mse_in, mse_out = tr.val_step(state, batch, masks=mk)
    lat = tr._fitted(state, batch[:, :3], mk)
    z0 = tuple(torch.tensor(lat[k].cpu().double().numpy()) for k in ("p_pos", "a", "gaussian_window"))
    rp = T.to_torch(oprm, torch.float64)
    sol = OT.solve_latent_ode(lambda z, _: OT.ponita_ode(rp, ocfg, z), z0, 0, 4, 1, "euler")
    p_fl, a_fl, w_fl = (v.reshape(10, *v.shape[2:]) for v in sol)
    rec = T.nef_apply(T.to_torch(prm, torch.float64), cfg, torch.tensor(coords)[None].expand(10, -1, -1), p_fl, a_fl, w_fl)
    err = (rec.reshape(2, 5, 8, 8, 1).numpy() - traj) ** 2
    assert abs(float(mse_in) - err[:, :3].mean()) < 1e-4 * err[:, :3].mean()
    assert abs(float(mse_out) - err[:, 3:].mean()) < 1e-4 * err[:, 3:].mean()
