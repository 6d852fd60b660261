"""ORACLE (test infrastructure, not product code) -- fp64 numpy restatement of the latent ODE models and solvers.

PARITY UNPINNED (same reasons as enf_ref_np.py: the reference ships no tests or vectors and JAX/Flax are not
installed).  Pinned only by agreement with the torch restatement (ode_ref_torch.py), the equivariance properties
asserted in tests/test_ode_oracle.py and finite-difference gradient checks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference files restated (paths relative to /root/reference/experiments/fitting):
  PODE = ode_models/ponita_ode_g.py      (PolynomialFeatures :15-26, ConvBlock :29-49, SepGconv :52-83,
                                          PonitaGen :86-195, PonitaODEGen :198-258)
  MLPO = ode_models/mlp_ode.py           (MLPODE :5-42)
  SOLV = trainers/trainer_utils/solvers.py (_euler_step_treemapped :69-83, _rk4_step_treemapped :86-105,
                                          _solve_latent_ode :108-162)
Flax parameter names: nn.Sequential children are `layers_<index>` (index counts the parameter-free entries too).
"""
import math

import numpy as np

from . import enf_ref_np as R


def sa_invariant_spec(name, num_in=2):
    """Self-attention invariant (INV/__init__.py:13-44): the cross-attention classes, except 'ponita' -> Ponita2D
    (INV/ponita.py:48-92: dim 3, both sides carry an orientation)."""
    s = R.invariant_spec(name, num_in)
    if name == "ponita":
        s = dict(s, dim=3)
    return s


sa_invariant = R.sa_invariant        # (p, p) invariants, axis 1 = receiver side (PODE:158; p already expanded for 'ponita')


def poly_features(x, degree):
    """PODE:22-26: [x, x(x)x, (x(x)x)(x)x, ...] -- degree+1 Kronecker powers, flattened and concatenated."""
    out = [x]
    for _ in range(degree):
        out.append(np.einsum("...i,...j->...ij", out[-1], x).reshape(*x.shape[:-1], -1))
    return np.concatenate(out, -1)


def num_poly_features(I, degree):
    return sum(I ** k for k in range(1, degree + 2))


def dense(x, p):
    y = x @ p["kernel"]
    return y + p["bias"] if "bias" in p else y


def ponita_gen(P, cfg, p, a):
    """PonitaGen.__call__ (PODE:143-195).  cfg: invariant, num_in, degree, kernel_size, vec_num_out, global_pool."""
    name = cfg["invariant"]
    spec = sa_invariant_spec(name, cfg.get("num_in", 2))
    zp, zo = spec["z_pos"], spec["z_ori"]
    if zo > 0:                                                      # PODE:152-155
        p = np.concatenate([p[..., :zp], np.cos(p[..., zp:]), np.sin(p[..., zp:])], -1)
    inv = sa_invariant(name, p)                                     # (B, Z, Z, I)
    kb = dense(poly_features(inv, cfg["degree"]), P["kernel_basis"]["layers_1"])
    kb = R.gelu(dense(R.gelu(kb), P["kernel_basis"]["layers_3"]))  # PODE:102-104
    if cfg.get("kernel_size", "global") != "global":                # PODE:162-164
        kb = kb * np.exp(-np.linalg.norm(p[:, :, None, :] - p[:, None, :, :], axis=-1) / cfg["kernel_size"])[..., None]
    a = dense(a, P["a_stem"])                                       # PODE:167
    for i in range(cfg["num_layers"]):                              # ConvBlock, PODE:42-49
        L = P[f"interaction_layers_{i}"]
        kern = dense(kb, L["conv"]["kernel"])                       # (B, R, S, C)
        x = np.einsum("bsc,brsc->brc", a, kern) + L["conv"]["bias"]  # SepGconv, PODE:75-83
        x = R.layer_norm(x, L["norm"])
        a = dense(R.gelu(dense(x, L["linear_1"])), L["linear_2"])
    scalar = dense(a, P["readout_scalar"]["layers_0"])              # PODE:174
    vec = None
    if cfg.get("vec_num_out", 1) > 0:                               # PODE:176-193
        rel = p[:, :, None, :zp] - p[:, None, :, :zp]
        inv_a = np.concatenate([inv, np.broadcast_to(a[:, None, :, :], inv.shape[:-1] + (a.shape[-1],))], -1)
        vec = (dense(inv_a, P["readout_vec_rel"]) * rel).mean(-2)
        if zo > 0:
            ori = np.broadcast_to(p[:, None, :, zp:], rel.shape)
            vec = vec + (dense(inv_a, P["readout_vec_ori"]) * ori).mean(-2)
    if cfg.get("global_pool", False):
        scalar = scalar.mean(1)
        vec = vec.mean(1) if vec is not None else None
    return scalar, vec


def ponita_ode(params, cfg, latents):
    """PonitaODEGen.__call__ (PODE:228-258): (dp/dt, da/dt, dwindow/dt = 0)."""
    p, a, window = latents
    spec = sa_invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    scalar, vec = ponita_gen(params["params"]["ponita"], cfg, p, a - 1)
    if spec["z_ori"] > 0:
        da, dp = scalar[..., :-1], np.concatenate([vec, scalar[..., -1:]], -1)
    else:
        da, dp = scalar, vec
    return dp, da, (np.zeros_like(window) if window is not None else None)


def mlp_ode(params, latents):
    """MLPODE.__call__ (MLPO:31-42)."""
    p, a, window = latents
    h = np.concatenate([p, a - 1], -1)
    out = []
    for net in ("mlp_p", "mlp_a"):
        x = h
        for i in (0, 2, 4):
            x = R.gelu(dense(x, params["params"][net][f"layers_{i}"]))
        out.append(dense(x, params["params"][net]["layers_6"]))
    return out[0], out[1], np.zeros_like(window)


def _axpy(x, h, k):
    return tuple(None if xi is None else xi + h * ki for xi, ki in zip(x, k))


def euler_step(f, x, t, h):                                         # SOLV:69-83
    return _axpy(x, h, f(x, t))


def rk4_step(f, x, t, h):                                           # SOLV:86-105
    k1 = f(x, t)
    k2 = f(_axpy(x, 0.5 * h, k1), t + 0.5 * h)
    k3 = f(_axpy(x, 0.5 * h, k2), t + 0.5 * h)
    k4 = f(_axpy(x, h, k3), t + h)
    return tuple(None if xi is None else xi + (h / 6.0) * (a1 + 2 * a2 + 2 * a3 + a4)
                 for xi, a1, a2, a3, a4 in zip(x, k1, k2, k3, k4))


def solve_latent_ode(f, latents, t0, tf, h, method="rk4"):
    """SOLV:108-162: returns (p, a, window) trajectories of shape (batch, num_steps + 1, ...)."""
    num_steps = int((tf - t0) / h)
    traj, t = [tuple(latents)], t0
    for _ in range(num_steps):
        if method == "rk4":
            traj.append(rk4_step(f, traj[-1], t, h))
        elif method == "euler":
            traj.append(euler_step(f, traj[-1], t, h))
        else:
            raise ValueError(f"Unknown method: {method}")
        t += h
    return tuple(np.stack([s[i] for s in traj], 1) for i in range(3))


# ---------------------------------------------------------------- parameter initialisation (flax defaults)
def _readout_init(rng, n_in, n_out, scale=1e-6):
    """variance_scaling(1e-6, 'fan_in', 'truncated_normal') (PODE:124,130,132); `scale` can be raised in tests so
    that the readouts are not numerically invisible."""
    return {"kernel": R._vs(rng, (n_in, n_out), scale, "fan_in", "trunc")}


def init_ponita_ode(seed, cfg, latent_dim, jitter=0.0, readout_scale=1e-6):
    """Parameter tree of PonitaODEGen.init (shapes from PODE:97-134; SepGconv kernel: chang_xavier_uniform :9-13)."""
    rng = np.random.default_rng(seed)
    spec = sa_invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    H, J, wf = cfg["num_hidden"], cfg["basis_dim"], cfg["widening_factor"]
    F = num_poly_features(spec["dim"], cfg["degree"])
    P = {"kernel_basis": {"layers_1": R._dense_init(rng, F, H), "layers_3": R._dense_init(rng, H, J)},
         "a_stem": {"kernel": R._vs(rng, (latent_dim, H), 1.0, "fan_in", "trunc")}}
    for i in range(cfg["num_layers"]):
        lim = math.sqrt(2.0 / (J + H) * J)
        P[f"interaction_layers_{i}"] = {
            "conv": {"kernel": {"kernel": rng.uniform(-lim, lim, (J, H))}, "bias": np.zeros(H)},
            "norm": R._ln_init(H, rng, jitter),
            "linear_1": R._dense_init(rng, H, wf * H), "linear_2": R._dense_init(rng, wf * H, H)}
    n_sc = latent_dim + (1 if spec["z_ori"] > 0 else 0)             # PODE:214-217
    P["readout_scalar"] = {"layers_0": _readout_init(rng, H, n_sc, readout_scale)}
    if cfg.get("vec_num_out", 1) > 0:
        P["readout_vec_rel"] = _readout_init(rng, spec["dim"] + H, cfg.get("vec_num_out", 1), readout_scale)
        if spec["z_ori"] > 0:
            P["readout_vec_ori"] = _readout_init(rng, spec["dim"] + H, cfg.get("vec_num_out", 1), readout_scale)
    if jitter:
        for k, v in _leaves(P):
            if k == "bias":
                v += jitter * rng.standard_normal(v.shape)
    return {"params": {"ponita": P}}


def init_mlp_ode(seed, num_hidden, p_dim, latent_dim, vec_num_out=1):
    rng = np.random.default_rng(seed)
    out = {}
    for net, n_out in (("mlp_p", 2 * vec_num_out), ("mlp_a", latent_dim)):
        dims = [p_dim + latent_dim, num_hidden, num_hidden, num_hidden, n_out]
        out[net] = {f"layers_{2 * i}": R._dense_init(rng, dims[i], dims[i + 1]) for i in range(4)}
    return {"params": out}


def _leaves(tree):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from _leaves(v)
        else:
            yield k, v
