"""ball / ball_lat cross-attention invariants (config_ihc.yaml: invariant_type 'ball', num_hidden 32, num_heads 3,
25 latents of width 32; invariant/ball.py:54-96, ball_lat.py:66-88).  Their latent-only components enter the kernels as
per-latent RFF phases (csrc/enf_layout.h: enf_inv_rows), so forward, latent gradients (through the rotation matrix and the
phases) and weight gradients are each checked against the oracle."""
import numpy as np
import pytest

from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef
from tests.test_gpu_backward import ref_grads, hip_grads, rel
from tests import test_gpu_weight_grads as WG

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("inv", ["ball", "ball_lat"])
@pytest.mark.parametrize("D,H,C,O,Z,N", [(64, 2, 16, 2, 9, 70), (32, 3, 32, 1, 25, 50), (64, 1, 8, 1, 4, 33)])
def test_ball_forward_backward(cuda, D, H, C, O, Z, N, inv, precision, pair_variant, bwd_variant):
    cfg = make_cfg(inv, D=D, H=H, C=C, O=O, freq=(0.2, 0.5))             # config_ihc.yaml's frequency multipliers
    prm = R.init_params(D + Z, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 3, N, Z, D + H)
    w = np.random.default_rng(1).standard_normal((3, N, O))
    ro, rp, ra, rs = ref_grads(prm, cfg, x, p, a, s, w)
    ho, gp, ga, gs = hip_grads(cuda, build_nef(cfg, precision), prm, x, p, a, s, w)
    tol_o, tol_g = (2e-5, 2e-4) if precision == "f32" else (3e-2, 7e-2)
    assert np.abs(ho - ro).max() / np.abs(ro).max() < tol_o
    assert rel(ga, ra) < tol_g and rel(gs, rs) < tol_g
    assert rel(gp, rp) < tol_g
    if precision == "f32":          # every pose component on its own: Euler angles (through R and the window), radius (phase)
        for i in range(4):
            if np.abs(rp[..., i]).max() > 0:
                assert rel(gp[..., i], rp[..., i]) < 1e-3, (inv, i)
            else:
                assert np.abs(gp[..., i]).max() == 0, (inv, i)       # ball_lat ignores gamma


@pytest.mark.parametrize("inv", ["ball", "ball_lat"])
def test_ball_weight_grads(cuda, inv):
    cfg = make_cfg(inv, D=32, H=3, C=8, O=1, freq=(0.2, 0.5))
    WG.check(cuda, cfg, B=2, N=50, Z=9, precision="f32", seed=8)


def test_ball_needs_64_wide_kernels(cuda):
    cfg = make_cfg("ball", D=128, H=2, C=8, O=1)
    prm = R.init_params(1, cfg)
    x, p, a, s = make_inputs(cfg, 1, 16, 4, 1)
    nef = build_nef(cfg, "f32")
    import torch
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    with pytest.raises(NotImplementedError):
        nef.apply(nef.load_params(prm, device=cuda), t(x), t(p), t(a), t(s))
