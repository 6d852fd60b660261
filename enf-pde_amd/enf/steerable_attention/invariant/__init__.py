"""Invariant descriptors + factories, mirroring enf/steerable_attention/invariant/__init__.py:13-78.

The arithmetic of each invariant and of its gaussian window lives in the HIP kernels
(csrc/enf_device.h: pair_invariant); these classes carry the metadata the reference's callers
read (``dim``, ``num_x_pos_dims``, ``num_z_pos_dims``, ``num_z_ori_dims``, ``is_periodic``;
_base_invariant.py:7-23, trainers/_base_pde_trainer.py:62-63) and the kernel id.
"""
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
        raise NotImplementedError("invariants are evaluated inside the fused HIP kernel; "
                                  "use EquivariantCrossAttentionNeF.apply")


def _mk(name_, dim, xpos, zpos, zori, periodic):
    class _Inv(BaseInvariant):
        name = name_

        def __init__(self, num_dims=None):
            super().__init__()
            nd = num_dims if num_dims is not None else 2
            self.dim = dim(nd)
            self.num_x_pos_dims = xpos(nd)
            self.num_x_ori_dims = 0
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
    """Self-attention invariant (invariant/__init__.py:13-44); only consulted when num_layers > 0."""
    return get_ca_invariant(cfg)
