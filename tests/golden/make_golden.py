"""Generates the golden fixtures in this directory from the oracle (fp64 numpy forward, fp64 torch
autograd gradients, one inner-loop trace).  The reference itself cannot be executed here (JAX/Flax
absent), so these vectors pin the ORACLE (regression) and give the HIP path a fixed target; they do
not pin the oracle to the reference ("parity unpinned", oracle/enf_ref_np.py).

    python tests/golden/make_golden.py [case ...]   # rewrites tests/golden/*.npz (all, or the named decoder cases)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import enf_ref_np as R          # noqa: E402
from oracle import enf_ref_torch as T       # noqa: E402
from tests.helpers import make_cfg, make_inputs   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (cfg kwargs, B, N, Z, param seed, store weights?)
CASES = {
    "tiny_rel_pos_periodic": (dict(invariant="rel_pos_periodic", D=64, H=1, C=8, O=2), 2, 7, 4, 11, True),
    "tiny_ponita": (dict(invariant="ponita", D=64, H=2, C=8, O=1, freq=(0.05, 0.2)), 2, 9, 4, 12, False),
    "tiny_polar_periodic": (dict(invariant="polar_periodic", D=64, H=2, C=4, O=1, freq=(0.5, 0.5)), 2, 11, 6, 13, False),
    "tiny_latitude_periodic": (dict(invariant="latitude_periodic", D=64, H=2, C=8, O=3, freq=(0.05, 0.2)), 1, 13, 8, 14, False),
    "cfg2_rel_pos_periodic": (dict(invariant="rel_pos_periodic", D=128, H=2, C=16, O=1), 2, 64, 64, 15, False),
    "tiny_ball": (dict(invariant="ball", D=64, H=2, C=8, O=1, freq=(0.2, 0.5)), 2, 9, 5, 16, False),          # config_ihc.yaml's invariant
    "tiny_ball_lat": (dict(invariant="ball_lat", D=64, H=1, C=8, O=2, freq=(0.2, 0.5)), 2, 7, 4, 17, False),
    # BASELINE config 3's decoder (config_shallow_water.yaml:39-55: latitude_periodic, 128 latents, latent_dim 32, 3 fields) WITH its
    # weights: the test loads them from the fixture, so it does not depend on init_params' random stream
    "cfg3_latitude_periodic": (dict(invariant="latitude_periodic", D=128, H=2, C=32, O=3, freq=(0.05, 0.2)), 1, 48, 128, 18, True),
}


def flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def round_f32(tree):
    """the tree with every leaf rounded to fp32 (what a fixture stores), still as float64 arrays"""
    return {k: round_f32(v) if isinstance(v, dict) else np.asarray(v, np.float32).astype(np.float64) for k, v in tree.items()}


def unflatten(g, prefix="W/"):
    """{'params': tree} from a fixture's 'W/a/b/c' entries (float64)"""
    tree = {}
    for key in g.files if hasattr(g, "files") else g:
        if not key.startswith(prefix):
            continue
        d = tree
        parts = key[len(prefix):].split("/")
        for part in parts[:-1]:
            d = d.setdefault(part, {})
        d[parts[-1]] = np.asarray(g[key], np.float64)
    return {"params": tree} if tree else None


def make_case(name):
    kw, B, N, Z, seed, store_w = CASES[name]
    cfg = make_cfg(**kw)
    prm = R.init_params(seed, cfg, jitter=0.1)
    if store_w and name != "tiny_rel_pos_periodic":      # (that first fixture keeps its fp64-weight outputs: regenerating is bit-stable)
        prm = {"params": round_f32(prm["params"])}       # outputs below belong to exactly the weights stored
    x, p, a, s = make_inputs(cfg, B, N, Z, seed + 100)
    w = np.random.default_rng(seed + 200).standard_normal((B, N, cfg["num_out"]))
    out = R.nef_apply(prm, cfg, x, p, a, s)
    tp = T.to_torch(prm, torch.float64)
    tpp, ta, ts = (torch.tensor(v, requires_grad=True) for v in (p, a, s))
    o2 = T.nef_apply(tp, cfg, torch.tensor(x), tpp, ta, ts)
    (o2 * torch.tensor(w)).sum().backward()
    assert np.abs(o2.detach().numpy() - out).max() < 1e-10
    rec = dict(x=x, p=p, a=a, sigma=s, w=w, out=out, dp=tpp.grad.numpy(), da=ta.grad.numpy(), dsigma=ts.grad.numpy(),
               param_seed=np.int64(seed), jitter=np.float64(0.1))
    if store_w:
        for k, v in flatten(prm["params"]).items():
            rec["W/" + k] = v.astype(np.float32)
    return cfg, rec


def make_inner_loop():
    """3-step inner-loop trace (explicit masks) at BASELINE config 1's shape family, fp64."""
    cfg = make_cfg(invariant="ponita", D=64, H=2, C=16, O=1, freq=(0.05, 0.01))
    prm = R.init_params(21, cfg, jitter=0.05)
    B, Z, S, Ns = 2, 4, 3, 48
    g = 12
    lin = np.linspace(-1, 1, g)
    coords = np.stack(np.meshgrid(lin, lin), -1).reshape(-1, 2)
    rng = np.random.default_rng(22)
    img = np.cos(np.pi * coords @ rng.standard_normal((2, B))).T[..., None] + 0.1 * rng.standard_normal((B, g * g, 1))
    masks = np.stack([rng.permutation(g * g)[:Ns] for _ in range(S + 1)], 1)
    lat = R.init_latents(1, Z, 16, "ponita")
    lrs = {"p_pos": np.array([1.0]), "p_ori": np.array([1.0]), "a": np.full(16, 5.0), "gaussian_window": np.array([0.0])}
    t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    loss, fitted = T.inner_loop(T.to_torch(prm, torch.float64), cfg, {k: t64(v) for k, v in lat.items()},
                                {k: t64(v) for k, v in lrs.items()}, t64(coords), t64(img), torch.tensor(masks))
    rec = dict(coords=coords, img=img, masks=masks, loss=np.float64(loss.item()), param_seed=np.int64(21), jitter=np.float64(0.05))
    for k, v in lat.items():
        rec["lat0/" + k] = v
    for k, v in lrs.items():
        rec["lr/" + k] = v
    for k, v in fitted.items():
        rec["fit/" + k] = v.detach().numpy()
    return rec


def make_config1_trace():
    """BASELINE.json config 1 at FULL size (32 x 32 grid, 16 latents, D = 64, H = 2, C = 16, ponita, N_s = 1024,
    3 inner SGD steps, B = 8; hparams config_diff_plane.yaml:39-55,92-96): the inner-loop trace and the decode of the fitted
    latents on the full grid, fp64 oracle."""
    cfg = make_cfg(invariant="ponita", D=64, H=2, C=16, O=1, freq=(0.05, 0.01))
    prm = R.init_params(41, cfg, jitter=0.05)
    B, Z, S, g = 8, 16, 3, 32
    lin = np.linspace(-1, 1, g)
    coords = np.stack(np.meshgrid(lin, lin), -1).reshape(-1, 2)
    rng = np.random.default_rng(42)
    ks = np.stack(np.meshgrid(np.arange(-4, 5), np.arange(-4, 5), indexing="ij"), -1).reshape(-1, 2)
    img = (rng.standard_normal((B, 1, len(ks))) * np.cos(np.pi * coords @ ks.T + rng.uniform(0, 2 * np.pi, (B, 1, len(ks))))).sum(-1)
    img = (img / img.std(1, keepdims=True))[..., None]                    # band-limited, unit variance
    masks = np.stack([rng.permutation(g * g) for _ in range(S + 1)], 1)     # N_s = N: every step sees a permutation of the grid
    lat = R.init_latents(1, Z, 16, "ponita")
    lrs = {"p_pos": np.array([1.0]), "p_ori": np.array([1.0]), "a": np.full(16, 5.0), "gaussian_window": np.array([0.0])}
    t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    tp = T.to_torch(prm, torch.float64)
    loss, fitted = T.inner_loop(tp, cfg, {k: t64(v) for k, v in lat.items()}, {k: t64(v) for k, v in lrs.items()}, t64(coords),
                                t64(img), torch.tensor(masks))
    fit = {k: v.detach() for k, v in fitted.items()}
    with torch.no_grad():
        recon = T.nef_apply(tp, cfg, t64(coords)[None].expand(B, -1, -1), torch.cat([fit["p_pos"], fit["p_ori"]], -1), fit["a"],
                            fit["gaussian_window"])
    rec = dict(coords=coords, img=img, masks=masks, loss=np.float64(loss.item()), recon=recon.numpy(), param_seed=np.int64(41),
               jitter=np.float64(0.05))
    for k, v in lat.items():
        rec["lat0/" + k] = v
    for k, v in lrs.items():
        rec["lr/" + k] = v
    for k, v in fit.items():
        rec["fit/" + k] = v.numpy()
    return rec


# latent-ODE fixtures: name -> (ode cfg kwargs, B, Z, latent_dim, param seed)
ODE_CASES = {
    "ode_rel_pos_periodic": (dict(invariant="rel_pos_periodic", num_hidden=32, basis_dim=16, num_layers=2), 2, 6, 8, 31),
    "ode_ponita": (dict(invariant="ponita", num_hidden=16, basis_dim=16, num_layers=3, kernel_size=0.2), 2, 5, 4, 32),
}


def make_ode_case(name):
    """PonitaODEGen derivative, its gradients for a fixed cotangent, and 4-step Euler / RK4 roll-outs (fp64 oracle)."""
    from oracle import ode_ref_np as O
    from oracle import ode_ref_torch as OT
    from tests.test_ode_oracle import ode_cfg, ode_inputs
    kw, B, Z, C, seed = ODE_CASES[name]
    cfg = ode_cfg(kw["invariant"], **{k: v for k, v in kw.items() if k != "invariant"})
    prm = O.init_ponita_ode(seed, cfg, latent_dim=C, jitter=0.1, readout_scale=0.05)
    p, a, w = ode_inputs(cfg, B, Z, C, seed + 100)
    rng = np.random.default_rng(seed + 200)
    wp, wa = rng.standard_normal(p.shape), rng.standard_normal(a.shape)
    dp, da, _ = O.ponita_ode(prm, cfg, (p, a, w))
    tp = T.to_torch(prm, torch.float64)
    tpp, ta = torch.tensor(p, requires_grad=True), torch.tensor(a, requires_grad=True)
    odp, oda, _ = OT.ponita_ode(tp, cfg, (tpp, ta, torch.tensor(w)))
    ((odp * torch.tensor(wp)).sum() + (oda * torch.tensor(wa)).sum()).backward()
    assert np.abs(odp.detach().numpy() - dp).max() < 1e-10
    rec = dict(p=p, a=a, window=w, wp=wp, wa=wa, dp=dp, da=da, gp=tpp.grad.numpy(), ga=ta.grad.numpy(),
               param_seed=np.int64(seed), jitter=np.float64(0.1), readout_scale=np.float64(0.05))
    for method in ("euler", "rk4"):
        tr = O.solve_latent_ode(lambda z, t: O.ponita_ode(prm, cfg, z), (p, a, w), 0, 4, 1, method=method)
        rec[method + "/p"], rec[method + "/a"] = tr[0], tr[1]
    return cfg, rec


if __name__ == "__main__":
    for name in ([] if sys.argv[1:] else ODE_CASES):
        cfg, rec = make_ode_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
        print(name, {k: v.shape for k, v in rec.items() if hasattr(v, "shape")})
    only = sys.argv[1:]
    for name in CASES:
        if only and name not in only:
            continue
        cfg, rec = make_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
        print(name, {k: v.shape for k, v in rec.items() if hasattr(v, "shape") and not k.startswith("W/")})
    if sys.argv[1:]:
        sys.exit(0)
    np.savez_compressed(os.path.join(HERE, "inner_loop_ponita.npz"), **make_inner_loop())
    print("inner_loop_ponita written")
    np.savez_compressed(os.path.join(HERE, "config1_trace.npz"), **make_config1_trace())
    print("config1_trace written")
