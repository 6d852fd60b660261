"""ORACLE (test infrastructure, not product code) -- un-fused PyTorch restatement with autograd.

PARITY UNPINNED (see enf_ref_np.py header): no reference fixtures exist and JAX is not
installed; this restatement is pinned by agreement with the independent numpy fp64
restatement, invariance properties and finite differences only.

Same tensor materialisation as the reference's jnp code (every (B,N,Z,.) intermediate is a
real tensor); autograd supplies d/d(p,a,sigma), d/d(weights) and grad-of-grad, which is what
the reference gets from jax.grad (experiments/fitting/trainers/pde_trainer.py:188,200,255).
It also serves as bench.py's ``cpu_baseline`` ("port": PyTorch-CPU restatement, JAX unavailable).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch

from .enf_ref_np import LN_EPS, invariant_spec


def to_torch(tree, dtype=torch.float32, device="cpu", requires_grad=False):
    if isinstance(tree, dict):
        return {k: to_torch(v, dtype, device, requires_grad) for k, v in tree.items()}
    t = torch.as_tensor(np.asarray(tree), dtype=dtype, device=device).clone()
    if requires_grad:
        t.requires_grad_(True)
    return t


def dense(x, p):
    return x @ p["kernel"] + p["bias"]


def layer_norm(x, p):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * p["scale"] + p["bias"]


def gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def pointwise_ffn(x, p):                                            # ECA:16-21
    return dense(layer_norm(gelu(dense(x, p["Dense_0"])), p["LayerNorm_0"]), p["Dense_1"])


def rff_net(inv, p):                                                # RFF:42-47,86-93
    coeff = p["encoding"]["coefficients"].detach()                  # stop_gradient, RFF:90
    proj = (2.0 * math.pi * inv) @ coeff
    h = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
    h = torch.relu(dense(h, p["layers_0"]["linear"]))
    return dense(h, p["linear_final"])


def _sph_unit(ang):
    phi, th = ang[..., 0], ang[..., 1]
    return torch.stack([torch.sin(th) * torch.cos(phi), torch.sin(th) * torch.sin(phi), torch.cos(th)], dim=-1)


def _sph_cos(x, p):
    xv, pv = _sph_unit(x), _sph_unit(p)
    dot = torch.einsum("bnd,bmd->bnm", xv, pv)[..., None]
    return dot / (xv.norm(dim=-1)[:, :, None, None] * pv.norm(dim=-1)[:, None, :, None])


def invariant(name, x, p):
    if name == "rel_pos_periodic":
        rel = p[:, None, :, :] - x[:, :, None, :]
        return torch.cat([torch.cos(math.pi * rel), torch.sin(math.pi * rel)], dim=-1)
    if name == "latitude_periodic":
        B, N, Z = x.shape[0], x.shape[1], p.shape[1]
        e = lambda t: t.expand(B, N, Z)[..., None]
        return torch.cat([e(x[:, :, None, 1]), e(p[:, None, :, 1]),
                          torch.cos(e(x[:, :, None, 0]) - e(p[:, None, :, 0])),
                          torch.sin(e(x[:, :, None, 0]) - e(p[:, None, :, 0]))], dim=-1)
    if name == "polar_periodic":
        return _sph_cos(x, p)
    if name == "ponita":
        rel = x[:, :, None, :] - p[:, None, :, :2]
        ori = p[:, None, :, 2:]
        i1 = rel[..., 0] * ori[..., 0] + rel[..., 1] * ori[..., 1]
        i2 = -rel[..., 0] * ori[..., 1] + rel[..., 1] * ori[..., 0]
        return torch.stack([i1, i2], dim=-1)
    if name == "abs_pos":
        return x[:, :, None, :].expand(x.shape[0], x.shape[1], p.shape[1], x.shape[2])
    if name == "rel_pos":
        return x[:, :, None, :] - p[:, None, :, :]
    if name == "norm_rel_pos":
        return (p[:, None, :, :] - x[:, :, None, :]).norm(dim=-1, keepdim=True)
    if name == "ball":                              # INV/ball.py:54-96
        B, N, Z = x.shape[0], x.shape[1], p.shape[1]
        xv = _sph_unit(x)
        al, be, ga = p[..., 0], p[..., 1], p[..., 2]
        ca, sa, cb, sb, cg, sg = torch.cos(al), torch.sin(al), torch.cos(be), torch.sin(be), torch.cos(ga), torch.sin(ga)
        R = torch.stack([torch.stack([ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg], dim=-1),
                         torch.stack([sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg], dim=-1),
                         torch.stack([-sb, cb * sg, cb * cg], dim=-1)], dim=-2)
        rot = torch.einsum("bzij,bnj->bnzi", R, xv)
        return torch.cat([rot, x[:, :, None, 2:3].expand(B, N, Z, 1), p[:, None, :, 3:4].expand(B, N, Z, 1)], dim=-1)
    if name == "ball_lat":                          # INV/ball_lat.py:66-88
        B, N, Z = x.shape[0], x.shape[1], p.shape[1]
        e = lambda t: t.expand(B, N, Z)[..., None]
        dphi = e(x[:, :, None, 0]) - e(p[:, None, :, 0])
        return torch.cat([e(x[:, :, None, 1]), e(p[:, None, :, 1]), torch.cos(dphi), torch.sin(dphi),
                          e(x[:, :, None, 2]), e(p[:, None, :, 3])], dim=-1)
    raise ValueError(f"Unknown invariant type: {name}.")


def gaussian_window(name, x, p, sigma, num_in=2):
    spec = invariant_spec(name, num_in)
    zp, xp, kind = spec["z_pos"], spec["dx"], spec["window"]
    if kind == "nonperiodic":
        d2 = ((p[:, None, :, :zp] - x[:, :, None, :xp]) ** 2).sum(-1, keepdim=True)
        return -(1.0 / sigma[:, None, :] ** 2) * d2
    if kind == "periodic":
        nrd = -(torch.cos(math.pi * (p[:, None, :, :zp] - x[:, :, None, :xp])) ** 2).sum(-1, keepdim=True)
        return -(1.0 / sigma[:, None, :] ** 2) * nrd
    if kind == "sphere":
        dist = torch.arccos(torch.clamp(_sph_cos(x, p), -1 + 1e-6, 1 - 1e-6))
        return torch.exp(-dist ** 2 / (2 * sigma[:, None, :, :] ** 2))
    raise ValueError(kind)


def sa_invariant(name, p):
    if name == "ponita":                            # Ponita2D, INV/ponita.py:64-92 (p = (pos, cos, sin))
        rel = p[:, :, None, :2] - p[:, None, :, :2]
        ox, op = p[:, :, None, 2:], p[:, None, :, 2:]
        return torch.stack([rel[..., 0] * op[..., 0] + rel[..., 1] * op[..., 1],
                            -rel[..., 0] * op[..., 1] + rel[..., 1] * op[..., 0], (ox * op).sum(-1)], -1)
    return invariant(name, p, p)


def cross_attention(pa, cfg, x, p, a, sigma, self_attn=False):
    H, D, name = cfg["num_heads"], cfg["num_hidden"], cfg["invariant"]
    inv = sa_invariant(name, p) if self_attn else invariant(name, x, p)
    q = dense(rff_net(inv, pa["invariant_embedding_query"]), pa["inv_emb_to_q"])
    k = dense(a, pa["a_to_k"])
    v = dense(a, pa["a_to_v"])
    if cfg.get("condition_value_transform", True):
        gb = pointwise_ffn(rff_net(inv, pa["invariant_embedding_value"]), pa["inv_emb_to_v"])
        gam, bet = torch.chunk(gb, 2, dim=-1)
        v = v[:, None, :, :] * (1 + gam) + bet
        v = v.reshape(v.shape[:-1] + (H, D))
        v = pointwise_ffn(v, pa["inv_emb_cond_mixer"])
    else:
        v = v[:, None, :, :]
        v = v.reshape(v.shape[:-1] + (H, D))
    q = q.reshape(q.shape[:-1] + (H, D))
    k = k.reshape(k.shape[:-1] + (H, D))
    att = (q * k[:, None]).sum(-1) * (1.0 / D ** 0.5)
    if cfg.get("use_gaussian_window", True):
        att = att + gaussian_window(name, x, p, sigma, cfg.get("num_in", 2))
    att = torch.softmax(att, dim=-2)
    y = (att[..., None] * v).sum(dim=2)
    y = y.reshape(y.shape[0], y.shape[1], H * D)
    return dense(y, pa["out_proj"])


def nef_apply(params, cfg, x, p, a, sigma):
    """NEF:204-235.  All tensors share one dtype/device."""
    P = params["params"]
    spec = invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    if spec["z_ori"] > 0:
        zp = spec["z_pos"]
        p = torch.cat([p[:, :, :zp], torch.cos(p[:, :, zp:]), torch.sin(p[:, :, zp:])], dim=-1)
    a = dense(a, P["latent_stem"])
    for i in range(cfg.get("num_layers", 0)):                       # NEF:223-226
        sb = P[f"self_attention_blocks_{i}"]
        a_attn = cross_attention(sb["attn"], cfg, p, p, layer_norm(a, sb["layer_norm_attn"]), sigma, self_attn=True)
        a = gelu(a + pointwise_ffn(a + a_attn, sb["pointwise_ffn"]))
    blk = P["cross_attention_blocks_0"]
    att = cross_attention(blk["attn"], cfg, x, p, layer_norm(a, blk["layer_norm_attn"]), sigma)
    out = gelu(pointwise_ffn(att, blk["pointwise_ffn"]))
    o = P["out_proj"]
    out = gelu(dense(out, o["layers_0"]))
    out = gelu(dense(out, o["layers_2"]))
    return dense(out, o["layers_4"])


def nef_apply_chunked(params, cfg, x, p, a, sigma, chunk=512):
    """Decode in query chunks like pde_trainer.py:397-402 (bounds the (B,N,Z,.) intermediates)."""
    outs = [nef_apply(params, cfg, x[:, i:i + chunk], p, a, sigma) for i in range(0, x.shape[1], chunk)]
    return torch.cat(outs, dim=1)


def split_pose(latents, spec):
    """ADM:8-25 -- p = cat(p_pos, p_ori)."""
    if spec["z_ori"] > 0:
        return torch.cat([latents["p_pos"], latents["p_ori"]], dim=-1)
    return latents["p_pos"]


def inner_loop(params, cfg, latents0, lrs, coords, img, masks, optimize_gaussian_window=False,
               create_graph=False):
    """MAML inner loop, pde_trainer.py:133-235, with the sampling masks passed explicitly.

    latents0: dict of (1,Z,.) meta-init tensors (keys p_pos, a, gaussian_window[, p_ori])
    lrs:      dict of inner learning rates, same keys (scalar for poses/window, (C,) for a)
    coords:   (N,dx) grid;  img: (B,N,O) targets;  masks: LongTensor (N_s, S+1)
    Returns (loss at step S on masks[:, S], dict of fitted latents (B,Z,.)).
    """
    spec = invariant_spec(cfg["invariant"], cfg.get("num_in", 2))
    B = img.shape[0]
    S = masks.shape[1] - 1
    lat = {k: v.repeat_interleave(B, dim=0) for k, v in latents0.items()}      # TR:157-159
    if not create_graph:
        lat = {k: v.detach().clone().requires_grad_(True) for k, v in lat.items()}

    def loss_fn(lat, s):
        xs = coords[masks[:, s]][None].expand(B, -1, -1)                        # TR:193-197
        ys = img[:, masks[:, s]]
        out = nef_apply(params, cfg, xs, split_pose(lat, spec), lat["a"], lat["gaussian_window"])
        return ((out - ys) ** 2).mean()                                         # TR:185

    for s in range(S):                                                          # TR:191
        keys = list(lat.keys())
        g = torch.autograd.grad(loss_fn(lat, s), [lat[k] for k in keys], create_graph=create_graph,
                                allow_unused=True)
        new = {}
        for k, gk in zip(keys, g):
            gk = torch.zeros_like(lat[k]) if gk is None else gk * B           # TR:207
            if k == "gaussian_window" and not optimize_gaussian_window:        # TR:210-212
                gk = torch.zeros_like(gk)
            new[k] = lat[k] - lrs[k] * gk                                      # TR:215-219
            if not create_graph:
                new[k] = new[k].detach().requires_grad_(True)
        lat = new
    return loss_fn(lat, S), lat                                                # TR:225-235
