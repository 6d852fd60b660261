"""Where does the residual error of the finite-difference meta-gradient come from?  The SAME adjoint recursion
(fitting/trainers/pde_trainer.py: meta_gradients) is run on the CPU oracle's decoder in fp64 and in fp32 and compared
with exact second-order autograd of the oracle: fp64 isolates the method (truncation / a missing term), fp32 adds the
rounding of first-order gradients.  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import numpy as np, torch
from oracle import enf_ref_torch as T
from tests.test_gpu_trainer import _problem, _oracle_meta_grads, _get
from enf_pde_amd.enf.models import TENSOR_PATHS
from enf_pde_amd.fitting.trainers import meta_gradients


class OracleNef:
    def __init__(self, cfg):
        self.cfg, self.use_gaussian_window = cfg, True
        self.cross_attn_invariant = NS(num_z_ori_dims=0)

    def param_tensors(self, params):
        return [_get(params["params"], p) for p in TENSOR_PATHS]

    def apply(self, params, x, p, a, s):
        return T.nef_apply(params, self.cfg, x, p, a, s)


for kw in (dict(), dict(B=8, Ns=64, side=8, Z=16)):
    cfg, prm, coords, img, lat0, lrs, masks = _problem(**kw)
    loss_r, gw_r, gl_r, gr_r = _oracle_meta_grads(cfg, prm, coords, img, lat0, lrs, masks)
    print("problem", kw)
    cols = {}
    for dt in (torch.float64, torch.float32):
        nef = OracleNef(cfg)
        params = T.to_torch(prm, dt)
        t = lambda v: torch.tensor(v, dtype=dt)
        for step in (2e-2, 1e-3):
            _, g = meta_gradients(nef, params, {k: t(v) for k, v in lat0.items()}, {k: t(v) for k, v in lrs.items()}, t(coords), t(img),
                                  torch.tensor(masks), second_order="fd", fd_step=step)
            cols[(str(dt)[-7:], step)] = g
    print("  columns:", list(cols))
    for i, path in enumerate(TENSOR_PATHS):
        nb = np.linalg.norm(gw_r[i])
        if nb == 0 or not ("layers_0" in path or path[-2:] in (("latent_stem", "kernel"), ("a_to_k", "kernel"), ("layers_4", "kernel"))):
            continue
        print(f"  {'/'.join(path[-3:]):48s}" + "  ".join(f"{np.linalg.norm(g['nef'][i].double().numpy() - gw_r[i]) / nb:8.1e}" for g in cols.values()))
    for k in ("p_pos", "a"):
        print(f"  lat0 {k:42s}" + "  ".join(f"{np.linalg.norm(g['autodecoder'][k].double().numpy() - gl_r[k]) / np.linalg.norm(gl_r[k]):8.1e}" for g in cols.values()))


# ---------------------------------------------------------------- the same with the relu masks frozen at phi_s
print("\nfrozen relu masks (oracle, fp64): relu linearised at the unperturbed latents in both perturbed passes")
import math
from enf_pde_amd.fitting.trainers import pde_trainer as PT
STATE = {"mode": None, "masks": [], "i": 0}
_orig_rff = T.rff_net


def rff_masked(inv, p):
    coeff = p["encoding"]["coefficients"].detach()
    proj = (2.0 * math.pi * inv) @ coeff
    h = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
    pre = T.dense(h, p["layers_0"]["linear"])
    if STATE["mode"] == "write":
        STATE["masks"].append((pre > 0).to(pre.dtype).detach())
        h = torch.relu(pre)
    elif STATE["mode"] == "read":
        m = STATE["masks"][STATE["i"]]
        STATE["i"] += 1
        h = pre * m.repeat(pre.shape[0] // m.shape[0], 1, 1, 1)
    else:
        h = torch.relu(pre)
    return T.dense(h, p["linear_final"])


T.rff_net = rff_masked
_orig_diff = PT._diff_grads


def diff_frozen(nef, weights, coords, img, masks, s, plus, minus, keys):
    base = {k: 0.5 * (plus[k] + minus[k]) for k in plus}              # phi_s
    STATE.update(mode="write", masks=[], i=0)
    with torch.no_grad():
        PT._loss(nef, PT._tree_from_tensors([w.detach() for w in weights]), coords, img, masks, s, base)
    STATE.update(mode="read", i=0)
    out = _orig_diff(nef, weights, coords, img, masks, s, plus, minus, keys)
    STATE.update(mode=None)
    return out


PT._diff_grads = diff_frozen
for kw in (dict(), dict(B=8, Ns=64, side=8, Z=16)):
    cfg, prm, coords, img, lat0, lrs, masks = _problem(**kw)
    loss_r, gw_r, gl_r, gr_r = _oracle_meta_grads(cfg, prm, coords, img, lat0, lrs, masks)
    nef, params = OracleNef(cfg), T.to_torch(prm, torch.float64)
    t = lambda v: torch.tensor(v, dtype=torch.float64)
    print("problem", kw, "steps 2e-2, 1e-3, 1e-4")
    cols = [meta_gradients(nef, params, {k: t(v) for k, v in lat0.items()}, {k: t(v) for k, v in lrs.items()}, t(coords), t(img),
                           torch.tensor(masks), second_order="fd", fd_step=step)[1] for step in (2e-2, 1e-3, 1e-4)]
    for i, path in enumerate(TENSOR_PATHS):
        nb = np.linalg.norm(gw_r[i])
        if nb == 0 or not ("layers_0" in path or path[-2:] in (("latent_stem", "kernel"), ("a_to_k", "kernel"), ("layers_4", "kernel"))):
            continue
        print(f"  {'/'.join(path[-3:]):48s}" + "  ".join(f"{np.linalg.norm(g['nef'][i].numpy() - gw_r[i]) / nb:8.1e}" for g in cols))
    for k in ("p_pos", "a"):
        print(f"  lat0 {k:42s}" + "  ".join(f"{np.linalg.norm(g['autodecoder'][k].numpy() - gl_r[k]) / np.linalg.norm(gl_r[k]):8.1e}" for g in cols))
