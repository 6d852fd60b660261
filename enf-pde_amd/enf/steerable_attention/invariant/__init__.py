"""Invariant descriptors + factories, mirroring enf/steerable_attention/invariant/__init__.py:13-78.

For the decoder (queries x latents) the arithmetic of each invariant and of its gaussian window lives in the HIP
kernels (csrc/enf_device.h: pair_invariant); these classes carry the metadata the reference's callers
read (``dim``, ``num_x_pos_dims``, ``num_z_pos_dims``, ``num_z_ori_dims``, ``is_periodic``;
_base_invariant.py:7-23, trainers/_base_pde_trainer.py:62-63) and the kernel id.  ``__call__(x, p)`` evaluates the
invariant as differentiable device tensor ops: that is the latents x latents use of the latent ODE
(ponita_ode_g.py:158, Z^2 pairs per signal), not the decoder's path.
"""
import math

import torch

from ...._lib import INVARIANT_IDS


class BaseInvariant:
    name = None

    def __init__(self):
        self.dim = None
        self.num_x_pos_dims = None
        self.num_x_ori_dims = None
        self.num_z_pos_dims = None
        self.num_z_ori_dims = None
        self.is_periodic = False

    @property
    def kernel_id(self):
        return INVARIANT_IDS[self.name]

    def __call__(self, x, p):
        """x (B, N, .), p (B, Z, .) -> (B, N, Z, dim)."""
        return _evaluate(self.name, x, p)


def _unit(phi, th):
    return torch.stack([th.sin() * phi.cos(), th.sin() * phi.sin(), th.cos()], -1)


def _evaluate(name, x, p):
    X, P = x[:, :, None, :], p[:, None, :, :]
    full = lambda t: t.expand(x.shape[0], x.shape[1], p.shape[1])
    if name == "rel_pos_periodic":                    # rel_pos_periodic.py:47-60
        d = math.pi * (P - X)
        return torch.cat([d.cos(), d.sin()], -1)
    if name == "rel_pos":                             # rel_pos.py:41
        return X - P
    if name == "abs_pos":                             # abs_pos.py:42
        return X.expand(-1, -1, p.shape[1], -1)
    if name == "norm_rel_pos":                        # norm_rel_pos.py:34
        return torch.linalg.norm(P - X, dim=-1, keepdim=True)
    if name in ("ponita", "ponita_full"):             # ponita.py:36-44 / :80-92 (p = (pos, cos, sin))
        rel, ori = X[..., :2] - P[..., :2], P[..., 2:4]
        out = [rel[..., 0] * ori[..., 0] + rel[..., 1] * ori[..., 1], -rel[..., 0] * ori[..., 1] + rel[..., 1] * ori[..., 0]]
        if name == "ponita_full":
            out.append((X[..., 2:4] * ori).sum(-1))
        return torch.stack(out, -1)
    if name == "polar_periodic":                      # polar_periodic.py:52-68
        return (_unit(X[..., 0], X[..., 1]) * _unit(P[..., 0], P[..., 1])).sum(-1, keepdim=True)
    if name == "latitude_periodic":                   # spherical_longitude.py:68-85
        d = X[..., 0] - P[..., 0]
        return torch.stack([full(X[..., 1]), full(P[..., 1]), d.cos(), d.sin()], -1)
    if name == "ball":                                # ball.py:54-96
        al, be, ga = p[..., 0], p[..., 1], p[..., 2]
        ca, sa, cb, sb, cg, sg = al.cos(), al.sin(), be.cos(), be.sin(), ga.cos(), ga.sin()
        R = torch.stack([torch.stack([ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg], -1),
                         torch.stack([sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg], -1),
                         torch.stack([-sb, cb * sg, cb * cg], -1)], -2)
        rot = torch.einsum("bzij,bnj->bnzi", R, _unit(x[..., 0], x[..., 1]))
        return torch.cat([rot, full(X[..., 2])[..., None], full(P[..., 3])[..., None]], -1)
    if name == "ball_lat":                            # ball_lat.py:66-88
        d = X[..., 0] - P[..., 0]
        return torch.stack([full(X[..., 1]), full(P[..., 1]), d.cos(), d.sin(), full(X[..., 2]), full(P[..., 3])], -1)
    raise ValueError(f"Unknown invariant type: {name}.")


def _mk(name_, dim, xpos, zpos, zori, periodic, xori=0):
    class _Inv(BaseInvariant):
        name = name_

        def __init__(self, num_dims=None):
            super().__init__()
            nd = num_dims if num_dims is not None else 2
            self.dim = dim(nd)
            self.num_x_pos_dims = xpos(nd)
            self.num_x_ori_dims = xori
            self.num_z_pos_dims = zpos(nd)
            self.num_z_ori_dims = zori
            self.is_periodic = periodic
    return _Inv


_c = lambda v: (lambda nd: v)
_n = lambda nd: nd
NormRelativePositionND = _mk("norm_rel_pos", _c(1), _n, _n, 0, False)            # norm_rel_pos.py:6-22
RelativePositionND = _mk("rel_pos", _n, _n, _n, 0, False)                        # rel_pos.py:4-24
AbsolutePositionND = _mk("abs_pos", _n, _n, _n, 0, False)                        # abs_pos.py:6-25
RelativePosition2DPeriodic = _mk("rel_pos_periodic", lambda nd: 2 * nd, _n, _n, 0, True)   # rel_pos_periodic.py:6-33
RelativePositionPolarPeriodic = _mk("polar_periodic", _c(1), _c(2), _c(2), 0, True)        # polar_periodic.py:6-33
RelativeLatitudePeriodic = _mk("latitude_periodic", _c(4), _c(2), _c(2), 0, True)          # spherical_longitude.py:6-32
PonitaPos2D = _mk("ponita", _c(2), _c(2), _c(2), 1, False)                                 # ponita.py:6-18
Ponita2D = _mk("ponita_full", _c(3), _c(2), _c(2), 1, False, xori=1)                                  # ponita.py:48-62 (self-attention / ODE only)
BallInvariant = _mk("ball", _c(5), _c(3), _c(4), 0, False)                                 # ball.py:6-33 (p = Euler angles + radius)
BallLatInvariant = _mk("ball_lat", _c(6), _c(3), _c(4), 0, False)                          # ball_lat.py:6-33
for _k, _v in list(globals().items()):
    if isinstance(_v, type) and issubclass(_v, BaseInvariant) and _v is not BaseInvariant:
        _v.__name__ = _v.__qualname__ = _k


def get_ca_invariant(cfg) -> BaseInvariant:
    """Cross-attention invariant from ``cfg.invariant_type`` / ``cfg.num_in`` (invariant/__init__.py:47-78)."""
    t = cfg.invariant_type
    if t == "norm_rel_pos":
        return NormRelativePositionND(num_dims=cfg.num_in)
    if t == "rel_pos":
        return RelativePositionND(num_dims=cfg.num_in)
    if t == "rel_pos_periodic":
        assert cfg.num_in == 2, "RelativePosition2DPeriodic currently only supports 2D input."
        return RelativePosition2DPeriodic(num_dims=cfg.num_in)
    if t == "ponita":
        assert cfg.num_in == 2, "Ponita2D currently only supports 2D input."
        return PonitaPos2D()
    if t == "abs_pos":
        return AbsolutePositionND(num_dims=cfg.num_in)
    if t == "polar_periodic":
        return RelativePositionPolarPeriodic()
    if t == "latitude_periodic":
        return RelativeLatitudePeriodic()
    if t == "ball":
        return BallInvariant()
    if t == "ball_lat":
        return BallLatInvariant()
    raise ValueError(f"Unknown invariant type: {t}.")


def get_sa_invariant(cfg) -> BaseInvariant:
    """Self-attention invariant (invariant/__init__.py:13-44): the cross-attention classes, except that 'ponita' gives the
    full Ponita2D (both sides carry an orientation).  Used by the latent ODE (fitting/ode_models)."""
    if cfg.invariant_type == "ponita":
        assert cfg.num_in == 2, "Ponita2D currently only supports 2D input."
        return Ponita2D()
    return get_ca_invariant(cfg)
