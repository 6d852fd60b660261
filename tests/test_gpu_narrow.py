"""Narrow models (num_hidden 16 / 32: config_diff_sphere.yaml uses 16) run zero-padded on the 64-wide kernels
(enf/models/_pad.py, EnfDesc.d_true).  The padding is exact, so the same tolerances as the native widths apply."""
import numpy as np
import pytest
import torch

from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef
from tests.test_gpu_backward import ref_grads, hip_grads, rel
from tests import test_gpu_weight_grads as WG

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("D,H,C,O,Z,inv", [(16, 2, 4, 1, 18, "polar_periodic"), (32, 2, 8, 2, 9, "rel_pos_periodic"),
                                           (16, 1, 4, 1, 8, "ponita"), (48, 2, 8, 1, 6, "rel_pos"),
                                           (32, 3, 32, 1, 25, "rel_pos"),       # config_ihc.yaml's width / heads / latents
                                           (64, 4, 8, 1, 7, "rel_pos_periodic"), (64, 3, 8, 2, 5, "ponita")])
def test_narrow_forward_backward(cuda, D, H, C, O, Z, inv, precision):
    cfg = make_cfg(inv, D=D, H=H, C=C, O=O, freq=(0.5, 1.0))
    prm = R.init_params(D + Z, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 3, 50, Z, D)
    w = np.random.default_rng(1).standard_normal((3, 50, O))
    ro, rp, ra, rs = ref_grads(prm, cfg, x, p, a, s, w)
    nef = build_nef(cfg, precision)
    assert nef._Dp == 64 and nef._Hp in (1, 2, 4)
    ho, gp, ga, gs = hip_grads(cuda, nef, prm, x, p, a, s, w)
    tol_o, tol_g = (2e-5, 2e-4) if precision == "f32" else (3e-2, 7e-2)
    assert np.abs(ho - ro).max() / np.abs(ro).max() < tol_o
    assert rel(ga, ra) < tol_g and rel(gs, rs) < tol_g
    assert (rel(gp, rp) if np.linalg.norm(rp) > 0 else np.abs(gp).max()) < tol_g


def test_narrow_weight_grads(cuda):
    cfg = make_cfg("polar_periodic", D=16, H=2, C=4, O=1, freq=(0.5, 1.0))
    WG.check(cuda, cfg, B=3, N=60, Z=18, precision="f32", seed=4)


def test_three_heads_weight_grads(cuda):
    cfg = make_cfg("rel_pos", D=32, H=3, C=8, O=1, freq=(0.5, 1.0))
    WG.check(cuda, cfg, B=2, N=50, Z=9, precision="f32", seed=6)
