"""ORACLE (test infrastructure, not product code) -- fp64 numpy restatement of the ENF decoder.

PARITY UNPINNED: the reference (david-knigge/enf-pde) ships no tests, golden vectors or
known-answer fixtures for this path, and JAX/Flax are not installed here (ordinary
ModuleNotFoundError), so the reference cannot be executed.  This file is an op-for-op
restatement, written from reading the reference as text; it materialises every
(B, N, Z, .) intermediate exactly as the reference's jnp code does.  It is pinned only by
(1) agreement with the independent torch restatement in ``enf_ref_torch.py``,
(2) the group-invariance properties the reference eyeballs in
    experiments/fitting/trainers/_base_pde_trainer.py:731-757, asserted numerically in tests/,
(3) finite-difference gradient checks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference files restated (paths relative to /root/reference):
  NEF = enf/models/equivariant_cross_attention_nef.py
  ECA = enf/steerable_attention/equivariant_cross_attention.py
  RFF = enf/steerable_attention/embedding/rff.py
  INV = enf/steerable_attention/invariant/*.py

Third-party semantics fixed explicitly (flax/jax versions are un-pinned in the reference,
README.md:31): Dense = x @ W(in,out) + b; LayerNorm = biased variance over the last axis,
eps 1e-6, then * scale + bias; gelu = tanh approximation (jax.nn.gelu default);
softmax = max-subtracted.
"""
import math

import numpy as np

LN_EPS = 1e-6  # flax.linen.LayerNorm default epsilon

# invariant name -> (I, dx, dp_raw, num_z_pos_dims, num_z_ori_dims, window kind)
# INV/__init__.py:47-78 (get_ca_invariant) picks the class; the per-class constructor sets dims.
INVARIANTS = {
    "rel_pos_periodic": dict(dim=4, dx=2, z_pos=2, z_ori=0, window="periodic"),      # INV/rel_pos_periodic.py:20-33
    "latitude_periodic": dict(dim=4, dx=2, z_pos=2, z_ori=0, window="sphere"),       # INV/spherical_longitude.py:19-32
    "polar_periodic": dict(dim=1, dx=2, z_pos=2, z_ori=0, window="sphere"),          # INV/polar_periodic.py:20-33
    "ponita": dict(dim=2, dx=2, z_pos=2, z_ori=1, window="nonperiodic"),             # INV/ponita.py:11-18
    "abs_pos": dict(dim=None, dx=None, z_pos=None, z_ori=0, window="nonperiodic"),   # INV/abs_pos.py:17-25 (dim = num_in)
    "rel_pos": dict(dim=None, dx=None, z_pos=None, z_ori=0, window="nonperiodic"),   # INV/rel_pos.py:17-24
    "norm_rel_pos": dict(dim=1, dx=None, z_pos=None, z_ori=0, window="nonperiodic"), # INV/norm_rel_pos.py:17-22
    "ball": dict(dim=5, dx=3, z_pos=4, z_ori=0, window="sphere"),                    # INV/ball.py:22-33 (p = Euler angles + radius)
    "ball_lat": dict(dim=6, dx=3, z_pos=4, z_ori=0, window="sphere"),                # INV/ball_lat.py:22-33
}


def invariant_spec(name, num_in=2):
    """Dimensions of a cross-attention invariant (INV/__init__.py:47-78)."""
    if name not in INVARIANTS:
        raise ValueError(f"Unknown invariant type: {name}.")  # INV/__init__.py:78
    s = dict(INVARIANTS[name])
    if s["dx"] is None:
        s["dx"] = num_in
    if s["z_pos"] is None:
        s["z_pos"] = num_in
    if s["dim"] is None:
        s["dim"] = num_in
    if name in ("rel_pos_periodic", "ponita") and num_in != 2:
        raise AssertionError(f"{name} currently only supports 2D input.")  # INV/__init__.py:62,65
    return s


# --------------------------------------------------------------------------- primitives
def dense(x, p):
    return x @ p["kernel"] + p["bias"]


def layer_norm(x, p):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS) * p["scale"] + p["bias"]


def gelu(x):
    # jax.nn.gelu(approximate=True)
    return 0.5 * x * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def relu(x):
    return np.maximum(x, 0.0)


def softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def pointwise_ffn(x, p):
    """ECA:16-21 -- Dense, gelu, LayerNorm (after the activation), Dense."""
    x = dense(x, p["Dense_0"])
    x = gelu(x)
    x = layer_norm(x, p["LayerNorm_0"])
    return dense(x, p["Dense_1"])


def rff_net(inv, p):
    """RFF:42-47 with num_layers=2 (EMB:25-27): encoding, one relu Layer, linear_final."""
    coeff = p["encoding"]["coefficients"]           # (I, D/2), no gradient (RFF:90)
    proj = (2.0 * np.pi * inv) @ coeff              # RFF:80,92
    h = np.concatenate([np.sin(proj), np.cos(proj)], axis=-1)   # RFF:84
    h = relu(dense(h, p["layers_0"]["linear"]))     # RFF:63-64
    return dense(h, p["linear_final"])              # RFF:46


# --------------------------------------------------------------------------- invariants
def _sph_unit(ang):
    phi, theta = ang[..., 0], ang[..., 1]
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=-1)


def _sph_cos(x, p):
    """INV/polar_periodic.py:52-68 -- normalised dot product of the unit vectors, (B,N,Z,1)."""
    xv, pv = _sph_unit(x), _sph_unit(p)
    dot = np.einsum("bnd,bmd->bnm", xv, pv)[..., None]
    nx = np.linalg.norm(xv, axis=-1)[:, :, None, None]
    npn = np.linalg.norm(pv, axis=-1)[:, None, :, None]
    return dot / (nx * npn)


def invariant(name, x, p):
    """(B,N,dx),(B,Z,dp) -> (B,N,Z,I).  `p` is already angle-embedded for ponita (NEF:214-217)."""
    if name == "rel_pos_periodic":                  # INV/rel_pos_periodic.py:47-60
        rel = p[:, None, :, :] - x[:, :, None, :]
        return np.concatenate([np.cos(np.pi * rel), np.sin(np.pi * rel)], axis=-1)
    if name == "latitude_periodic":                 # INV/spherical_longitude.py:68-85
        B, N, Z = x.shape[0], x.shape[1], p.shape[1]
        phi_x = np.broadcast_to(x[:, :, None, 0], (B, N, Z))[..., None]
        th_x = np.broadcast_to(x[:, :, None, 1], (B, N, Z))[..., None]
        phi_p = np.broadcast_to(p[:, None, :, 0], (B, N, Z))[..., None]
        th_p = np.broadcast_to(p[:, None, :, 1], (B, N, Z))[..., None]
        return np.concatenate([th_x, th_p, np.cos(phi_x - phi_p), np.sin(phi_x - phi_p)], axis=-1)
    if name == "polar_periodic":                    # INV/polar_periodic.py:40-68 (cosine, no acos)
        return _sph_cos(x, p)
    if name == "ponita":                            # INV/ponita.py:30-44
        x_pos = x[:, :, None, :]
        p_pos, p_ori = p[:, None, :, :2], p[:, None, :, 2:]
        rel = x_pos - p_pos
        i1 = rel[..., 0] * p_ori[..., 0] + rel[..., 1] * p_ori[..., 1]
        i2 = -rel[..., 0] * p_ori[..., 1] + rel[..., 1] * p_ori[..., 0]
        return np.stack([i1, i2], axis=-1)
    if name == "abs_pos":                           # INV/abs_pos.py:42
        return np.broadcast_to(x[:, :, None, :], (x.shape[0], x.shape[1], p.shape[1], x.shape[2]))
    if name == "rel_pos":                           # INV/rel_pos.py:41
        return x[:, :, None, :] - p[:, None, :, :]
    if name == "norm_rel_pos":                      # INV/norm_rel_pos.py:34
        return np.linalg.norm(p[:, None, :, :] - x[:, :, None, :], axis=-1, keepdims=True)
    if name == "ball":                              # INV/ball.py:54-96: [R(alpha,beta,gamma) x^, r_x, r_p]
        B, N, Z = x.shape[0], x.shape[1], p.shape[1]
        xv = _sph_unit(x)                                                       # (B,N,3)
        R = ball_rotation(p)                                                    # (B,Z,3,3)
        rot = np.einsum("bzij,bnj->bnzi", R, xv)
        r_x = np.broadcast_to(x[:, :, None, 2:3], (B, N, Z, 1))
        r_p = np.broadcast_to(p[:, None, :, 3:4], (B, N, Z, 1))
        return np.concatenate([rot, r_x, r_p], axis=-1)
    if name == "ball_lat":                          # INV/ball_lat.py:66-88
        B, N, Z = x.shape[0], x.shape[1], p.shape[1]
        e = lambda t: np.broadcast_to(t, (B, N, Z))[..., None]
        dphi = e(x[:, :, None, 0]) - e(p[:, None, :, 0])
        return np.concatenate([e(x[:, :, None, 1]), e(p[:, None, :, 1]), np.cos(dphi), np.sin(dphi),
                               e(x[:, :, None, 2]), e(p[:, None, :, 3])], axis=-1)
    raise ValueError(f"Unknown invariant type: {name}.")


def ball_rotation(p):
    """INV/ball.py:76-84: the 3x3 matrix built from the Euler angles (alpha, beta, gamma) = p[..., :3]."""
    al, be, ga = p[..., 0], p[..., 1], p[..., 2]
    ca, sa, cb, sb, cg, sg = np.cos(al), np.sin(al), np.cos(be), np.sin(be), np.cos(ga), np.sin(ga)
    return np.stack([np.stack([ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg], axis=-1),
                     np.stack([sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg], axis=-1),
                     np.stack([-sb, cb * sg, cb * cg], axis=-1)], axis=-2)


def gaussian_window(name, x, p, sigma, num_in=2):
    """Additive logit term (B,N,Z,1).  Quirks kept verbatim."""
    spec = invariant_spec(name, num_in)
    kind = spec["window"]
    zp, xp = spec["z_pos"], spec["dx"]
    if kind == "nonperiodic":                       # INV/_base_invariant.py:25-33
        d2 = ((p[:, None, :, :zp] - x[:, :, None, :xp]) ** 2).sum(-1, keepdims=True)
        return -(1.0 / sigma[:, None, :] ** 2) * d2
    if kind == "periodic":                          # INV/_base_invariant.py:35-43 (positive, period 1)
        nrd = -(np.cos(np.pi * (p[:, None, :, :zp] - x[:, :, None, :xp])) ** 2).sum(-1, keepdims=True)
        return -(1.0 / sigma[:, None, :] ** 2) * nrd
    if kind == "sphere":                            # INV/spherical_longitude.py:34-55, INV/polar_periodic.py:35-38
        ang = _sph_cos(x, p)
        dist = np.arccos(np.clip(ang, -1 + 1e-6, 1 - 1e-6))
        return np.exp(-dist ** 2 / (2 * sigma[:, None, :, :] ** 2))
    raise ValueError(kind)


# --------------------------------------------------------------------------- operator + model
def sa_invariant(name, p):
    """Self-attention invariant of (p, p) (INV/__init__.py:13-44): the cross-attention classes, except 'ponita' ->
    Ponita2D (INV/ponita.py:64-92: both sides carry an orientation; p = (pos, cos, sin)).  Axis 1 = the 'x' side."""
    if name == "ponita":
        rel = p[:, :, None, :2] - p[:, None, :, :2]
        ori_x, ori_p = p[:, :, None, 2:], p[:, None, :, 2:]
        i1 = rel[..., 0] * ori_p[..., 0] + rel[..., 1] * ori_p[..., 1]
        i2 = -rel[..., 0] * ori_p[..., 1] + rel[..., 1] * ori_p[..., 0]
        return np.stack([i1, i2, (ori_x * ori_p).sum(-1)], -1)
    return invariant(name, p, p)


def cross_attention(pa, cfg, x, p, a, sigma, return_aux=False, self_attn=False):
    """ECA:74-151 with condition_value_transform=True, condition_invariant_embedding=False (NEF:101-126);
    project_heads only changes out_proj's width (ECA:69-72).  self_attn: x is p and the invariant is the
    self-attention one (NEF:223-226)."""
    H, D = cfg["num_heads"], cfg["num_hidden"]
    name = cfg["invariant"]
    inv = sa_invariant(name, p) if self_attn else invariant(name, x, p)   # ECA:86
    emb_q = rff_net(inv, pa["invariant_embedding_query"])          # ECA:89
    q = dense(emb_q, pa["inv_emb_to_q"])                           # ECA:92  (B,N,Z,HD)
    k = dense(a, pa["a_to_k"])                                     # ECA:93  (B,Z,HD)
    v = dense(a, pa["a_to_v"])                                     # ECA:94
    if cfg.get("condition_value_transform", True):
        emb_v = rff_net(inv, pa["invariant_embedding_value"])      # ECA:100
        gb = pointwise_ffn(emb_v, pa["inv_emb_to_v"])              # ECA:112 (B,N,Z,2HD)
        gam, bet = np.split(gb, 2, axis=-1)                        # ECA:115
        v = v[:, None, :, :] * (1 + gam) + bet                     # ECA:118
        v = v.reshape(v.shape[:-1] + (H, D))                       # ECA:121
        v = pointwise_ffn(v, pa["inv_emb_cond_mixer"])             # ECA:122
    else:
        v = v[:, None, :, :]
        v = v.reshape(v.shape[:-1] + (H, D))                       # ECA:124-127
    q = q.reshape(q.shape[:-1] + (H, D))                           # ECA:130
    k = k.reshape(k.shape[:-1] + (H, D))                           # ECA:131
    att = (q * k[:, None, ...]).sum(-1) * (1.0 / D ** 0.5)         # ECA:59,134 (B,N,Z,H)
    if cfg.get("use_gaussian_window", True):
        att = att + gaussian_window(name, x, p, sigma, cfg.get("num_in", 2))   # ECA:137-139
    att = softmax(att, axis=-2)                                    # ECA:141 softmax over Z
    y = (att[..., None] * v).sum(axis=2)                           # ECA:144 (B,N,H,D)
    y = y.reshape(y.shape[0], y.shape[1], H * D)                   # ECA:147
    out = dense(y, pa["out_proj"])                                 # ECA:150
    if return_aux:
        return out, dict(inv=inv, att=att, y=y)
    return out


def nef_apply(params, cfg, x, p, a, sigma):
    """EquivariantCrossAttentionNeF.__call__ (NEF:204-235).

    x (B,N,dx)  p (B,Z,z_pos+z_ori)  a (B,Z,C)  sigma (B,Z,1)  ->  (B,N,O)
    """
    P = params["params"]
    spec = invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    x, p, a, sigma = (np.asarray(t, dtype=np.float64) for t in (x, p, a, sigma))
    if spec["z_ori"] > 0:                                          # NEF:214-217
        zp = spec["z_pos"]
        p = np.concatenate([p[:, :, :zp], np.cos(p[:, :, zp:]), np.sin(p[:, :, zp:])], axis=-1)
    a = dense(a, P["latent_stem"])                                 # NEF:220
    for i in range(cfg.get("num_layers", 0)):                      # NEF:223-226, block NEF:43-68 (residual=True)
        sb = P[f"self_attention_blocks_{i}"]
        a_attn = cross_attention(sb["attn"], cfg, p, p, layer_norm(a, sb["layer_norm_attn"]), sigma, self_attn=True)
        a = gelu(a + pointwise_ffn(a + a_attn, sb["pointwise_ffn"]))
    blk = P["cross_attention_blocks_0"]
    a_norm = layer_norm(a, blk["layer_norm_attn"])                 # NEF:56
    att = cross_attention(blk["attn"], cfg, x, p, a_norm, sigma)   # NEF:59
    out = pointwise_ffn(att, blk["pointwise_ffn"])                 # NEF:66 (residual=False)
    out = gelu(out)                                                # NEF:230
    o = P["out_proj"]                                              # NEF:196-202,233
    out = gelu(dense(out, o["layers_0"]))
    out = gelu(dense(out, o["layers_2"]))
    return dense(out, o["layers_4"])


# --------------------------------------------------------------------------- parameter init
def _vs(rng, shape, scale, mode, dist):
    fan_in = shape[0]
    var = scale / fan_in
    if dist == "normal":
        return rng.standard_normal(shape) * math.sqrt(var)
    if dist == "trunc":  # flax lecun_normal: truncated normal, stddev corrected by .87962566
        std = math.sqrt(var) / 0.87962566103423978
        v = rng.standard_normal(shape)
        bad = np.abs(v) > 2
        while bad.any():
            v[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(v) > 2
        return v * std
    if dist == "uniform":
        lim = math.sqrt(3 * var)
        return rng.uniform(-lim, lim, shape)
    raise ValueError(dist)


def _dense_init(rng, n_in, n_out, kind="lecun"):
    if kind == "lecun":            # flax Dense default: lecun_normal kernel, zeros bias
        return {"kernel": _vs(rng, (n_in, n_out), 1.0, "fan_in", "trunc"), "bias": np.zeros(n_out)}
    if kind == "rff_layer":        # RFF:55-60
        return {"kernel": _vs(rng, (n_in, n_out), 2.0, "fan_in", "normal"), "bias": rng.standard_normal(n_out) * 1e-6}
    if kind == "rff_final":        # RFF:35-40
        return {"kernel": _vs(rng, (n_in, n_out), 2.0, "fan_in", "uniform"), "bias": rng.standard_normal(n_out) * 1e-6}
    raise ValueError(kind)


def _ln_init(n, rng=None, jitter=0.0):
    s, b = np.ones(n), np.zeros(n)
    if jitter:
        s = s + jitter * rng.standard_normal(n)
        b = b + jitter * rng.standard_normal(n)
    return {"scale": s, "bias": b}


def _ffn_init(rng, n_in, n_hid, n_out, jitter):
    return {"Dense_0": _dense_init(rng, n_in, n_hid), "LayerNorm_0": _ln_init(n_hid, rng, jitter),
            "Dense_1": _dense_init(rng, n_hid, n_out)}


def _rff_init(rng, I, D, std):
    return {"encoding": {"coefficients": rng.standard_normal((I, D // 2)) * std},   # RFF:83
            "layers_0": {"linear": _dense_init(rng, D, D, "rff_layer")},
            "linear_final": _dense_init(rng, D, D, "rff_final")}


def init_params(seed, cfg, jitter=0.0):
    """Random weights with the reference's initialiser distributions (SURVEY 8a 'Weights').

    `jitter` perturbs biases / LayerNorm scale+bias away from their 0/1 init so parity tests
    exercise those terms (a trained checkpoint has non-trivial values there)."""
    rng = np.random.default_rng(seed)
    D, H, C, O = cfg["num_hidden"], cfg["num_heads"], cfg["latent_dim"], cfg["num_out"]
    spec = invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    I = spec["dim"]
    assert D % 2 == 0, "For the Fourier Features hidden_dim should be even"  # RFF:75-77
    fq, fv = cfg["embedding_freq_multiplier"]
    HD = H * D
    def attn_init(I_, n_proj):
        return {
            "invariant_embedding_query": _rff_init(rng, I_, D, fq),
            "invariant_embedding_value": _rff_init(rng, I_, D, fv),
            "inv_emb_to_q": _dense_init(rng, D, HD), "a_to_k": _dense_init(rng, D, HD),
            "a_to_v": _dense_init(rng, D, HD),
            "inv_emb_to_v": _ffn_init(rng, D, D, 2 * HD, jitter),
            "inv_emb_cond_mixer": _ffn_init(rng, D, D, D, jitter),
            "out_proj": _dense_init(rng, HD, n_proj),
        }
    attn = attn_init(I, HD)
    P = {
        "latent_stem": _dense_init(rng, C, D),
        "cross_attention_blocks_0": {"layer_norm_attn": _ln_init(D, rng, jitter), "attn": attn,
                                     "pointwise_ffn": _ffn_init(rng, HD, HD, HD, jitter)},
        "out_proj": {"layers_0": _dense_init(rng, HD, D), "layers_2": _dense_init(rng, D, D),
                     "layers_4": _dense_init(rng, D, O)},
    }
    I_sa = 3 if cfg["invariant"] == "ponita" else I                # Ponita2D (INV/ponita.py:61)
    for i in range(cfg.get("num_layers", 0)):                      # NEF:137-167: project_heads=True -> widths D
        P[f"self_attention_blocks_{i}"] = {"layer_norm_attn": _ln_init(D, rng, jitter), "attn": attn_init(I_sa, D),
                                           "pointwise_ffn": _ffn_init(rng, D, D, D, jitter)}
    if jitter:
        def jit(d):
            for k, v in d.items():
                if isinstance(v, dict):
                    jit(v)
                elif k == "bias" and v.ndim == 1:
                    d[k] = v + jitter * rng.standard_normal(v.shape)
        jit(P)
    return {"params": P}


def count_params(params):
    n = 0
    for v in params.values():
        n += count_params(v) if isinstance(v, dict) else int(np.prod(v.shape))
    return n


# --------------------------------------------------------------------------- latent init
def init_positions_grid(num_signals, num_latents, num_dims):
    """enf/latents/utils.py:73-103."""
    k = int(round(num_latents ** (1.0 / num_dims)))
    assert abs(round(num_latents ** (1.0 / num_dims), 5) % 1) < 1e-5, \
        "num_latents must be a power of the number of position dimensions"
    ax = np.linspace(-1 + 1 / k, 1 - 1 / k, k)
    g = np.stack(np.meshgrid(*[ax] * num_dims, indexing="ij"), axis=-1).reshape(-1, num_dims)
    return np.repeat(g[None], num_signals, axis=0)


def init_positions_polar(num_signals, num_latents, num_dims=2):
    """enf/latents/utils.py:36-70."""
    n = num_latents // 2
    assert abs(round(n ** (1.0 / num_dims), 5) % 1) < 1e-5
    k = int(round(n ** (1.0 / num_dims)))
    gphi = np.linspace(np.pi / (2 * k), 2 * np.pi - np.pi / (2 * k), 2 * k)
    gth = np.linspace((np.pi / 2) / k, np.pi - (np.pi / 2) / k, k)
    g = np.stack(np.meshgrid(gphi, gth, indexing="ij"), axis=-1).reshape(-1, num_dims)
    return np.repeat(g[None], num_signals, axis=0)


def init_latents(num_signals, num_latents, latent_dim, invariant_name, coordinate_system="cartesian", num_in=2):
    """enf/latents/autodecoder.py:18-56 -> dict {'p_pos','a','gaussian_window'[, 'p_ori']}."""
    spec = invariant_spec(invariant_name, num_in)
    d = spec["z_pos"]
    out = {}
    if coordinate_system == "cartesian":
        out["p_pos"] = init_positions_grid(num_signals, num_latents, d)
        k = int(round(num_latents ** (1.0 / d), 5))
        gw = d / k                                                 # AD:39-43
    elif coordinate_system == "polar":
        out["p_pos"] = init_positions_polar(num_signals, num_latents, d)
        k = int(round((num_latents // 2) ** (1.0 / d), 5))
        gw = d * np.pi / k                                         # AD:46-51
    elif coordinate_system == "ball":                              # LU:4-33, AD:53-54 (Euler angles on a Fibonacci lattice + radius)
        i = np.arange(1, num_latents + 1)
        pos = np.stack([np.arccos(1 - 2 * i / (num_latents + 1)), np.pi * (1 + 5 ** 0.5) * i,
                        np.arange(num_latents) * (2 * np.pi / num_latents), np.full(num_latents, 0.75)], -1)
        out["p_pos"] = np.repeat(pos[None], num_signals, axis=0)
        gw = 1.0
    else:
        raise ValueError(coordinate_system)
    if spec["z_ori"] > 0:                                          # AD:27-29, LU:106-109
        pos = init_positions_grid(num_signals, num_latents, d)
        out["p_ori"] = np.arctan2(pos[:, :, 0], pos[:, :, 1])[:, :, None]
    out["a"] = np.ones((num_signals, num_latents, latent_dim))     # AD:33
    out["gaussian_window"] = np.full((num_signals, num_latents, 1), gw)   # AD:56
    return out
