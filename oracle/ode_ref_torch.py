"""ORACLE (test infrastructure, not product code) -- PyTorch restatement (autograd) of the latent ODE models.

PARITY UNPINNED (see ode_ref_np.py): pinned by agreement with the numpy restatement, equivariance properties and
finite differences only.  Supplies d/d(p, a) and d/d(weights) of PonitaODEGen / MLPODE and of a solver roll-out,
which is what the reference gets from jax.grad over `ode_loss` (experiments/fitting/trainers/pde_trainer.py:411-500).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch

from . import enf_ref_torch as T
from .ode_ref_np import sa_invariant_spec


sa_invariant = T.sa_invariant


def poly_features(x, degree):                       # PODE:22-26
    out = [x]
    for _ in range(degree):
        out.append(torch.einsum("...i,...j->...ij", out[-1], x).reshape(*x.shape[:-1], -1))
    return torch.cat(out, -1)


def dense(x, p):
    y = x @ p["kernel"]
    return y + p["bias"] if "bias" in p else y


def ponita_gen(P, cfg, p, a):                       # PODE:143-195
    name = cfg["invariant"]
    spec = sa_invariant_spec(name, cfg.get("num_in", 2))
    zp, zo = spec["z_pos"], spec["z_ori"]
    if zo > 0:
        p = torch.cat([p[..., :zp], torch.cos(p[..., zp:]), torch.sin(p[..., zp:])], -1)
    inv = sa_invariant(name, p)
    kb = dense(poly_features(inv, cfg["degree"]), P["kernel_basis"]["layers_1"])
    kb = T.gelu(dense(T.gelu(kb), P["kernel_basis"]["layers_3"]))
    if cfg.get("kernel_size", "global") != "global":
        kb = kb * torch.exp(-torch.linalg.norm(p[:, :, None, :] - p[:, None, :, :], dim=-1) / cfg["kernel_size"])[..., None]
    a = dense(a, P["a_stem"])
    for i in range(cfg["num_layers"]):
        L = P[f"interaction_layers_{i}"]
        kern = dense(kb, L["conv"]["kernel"])
        x = torch.einsum("bsc,brsc->brc", a, kern) + L["conv"]["bias"]
        x = T.layer_norm(x, L["norm"])
        a = dense(T.gelu(dense(x, L["linear_1"])), L["linear_2"])
    scalar = dense(a, P["readout_scalar"]["layers_0"])
    vec = None
    if cfg.get("vec_num_out", 1) > 0:
        rel = p[:, :, None, :zp] - p[:, None, :, :zp]
        inv_a = torch.cat([inv, a[:, None, :, :].expand(*inv.shape[:-1], a.shape[-1])], -1)
        vec = (dense(inv_a, P["readout_vec_rel"]) * rel).mean(-2)
        if zo > 0:
            ori = p[:, None, :, zp:].expand_as(rel)
            vec = vec + (dense(inv_a, P["readout_vec_ori"]) * ori).mean(-2)
    if cfg.get("global_pool", False):
        scalar = scalar.mean(1)
        vec = vec.mean(1) if vec is not None else None
    return scalar, vec


def ponita_ode(params, cfg, latents):               # PODE:228-258
    p, a, window = latents
    spec = sa_invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    scalar, vec = ponita_gen(params["params"]["ponita"], cfg, p, a - 1)
    if spec["z_ori"] > 0:
        da, dp = scalar[..., :-1], torch.cat([vec, scalar[..., -1:]], -1)
    else:
        da, dp = scalar, vec
    return dp, da, (torch.zeros_like(window) if window is not None else None)


def mlp_ode(params, latents):                       # MLPO:31-42
    p, a, window = latents
    h = torch.cat([p, a - 1], -1)
    out = []
    for net in ("mlp_p", "mlp_a"):
        x = h
        for i in (0, 2, 4):
            x = T.gelu(dense(x, params["params"][net][f"layers_{i}"]))
        out.append(dense(x, params["params"][net]["layers_6"]))
    return out[0], out[1], torch.zeros_like(window)


def _axpy(x, h, k):
    return tuple(None if xi is None else xi + h * ki for xi, ki in zip(x, k))


def solve_latent_ode(f, latents, t0, tf, h, method="rk4"):          # SOLV:108-162
    num_steps = int((tf - t0) / h)
    traj, t = [tuple(latents)], t0
    for _ in range(num_steps):
        x = traj[-1]
        if method == "euler":
            traj.append(_axpy(x, h, f(x, t)))
        elif method == "rk4":
            k1 = f(x, t)
            k2 = f(_axpy(x, 0.5 * h, k1), t + 0.5 * h)
            k3 = f(_axpy(x, 0.5 * h, k2), t + 0.5 * h)
            k4 = f(_axpy(x, h, k3), t + h)
            traj.append(tuple(None if xi is None else xi + (h / 6.0) * (a1 + 2 * a2 + 2 * a3 + a4)
                              for xi, a1, a2, a3, a4 in zip(x, k1, k2, k3, k4)))
        else:
            raise ValueError(f"Unknown method: {method}")
        t += h
    return tuple(torch.stack([s[i] for s in traj], 1) for i in range(3))
