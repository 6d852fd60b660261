"""Per-tensor error of the meta-gradient (fd / first-order) vs exact second-order autograd of the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_trainer import _problem, _oracle_meta_grads
from tests.helpers import build_nef
from enf_pde_amd.enf.models import TENSOR_PATHS
from enf_pde_amd.fitting.trainers import meta_gradients
cuda = torch.device("cuda:0")
for kw in (dict(), dict(B=8, Ns=64, side=8, Z=16)):
    cfg, prm, coords, img, lat0, lrs, masks = _problem(**kw)
    loss_r, gw_r, gl_r, gr_r = _oracle_meta_grads(cfg, prm, coords, img, lat0, lrs, masks)
    nef = build_nef(cfg, "f32"); params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    res = {}
    # columns: first-order | fd, masks free, step 2e-2 | fd, relu masks frozen at phi_s, steps 2e-2, 5e-3, 1e-3
    for mode, step, frz in (("none", 0, False), ("fd", 2e-2, False), ("fd", 2e-2, True), ("fd", 5e-3, True), ("fd", 1e-3, True)):
        _, g = meta_gradients(nef, params, {k: t(v) for k, v in lat0.items()}, {k: t(v) for k, v in lrs.items()}, t(coords), t(img),
                              torch.tensor(masks, device=cuda), second_order=mode, fd_step=step or 5e-3, freeze_relu=frz)
        res[(mode, step, frz)] = g
    print("problem", kw)
    for i, path in enumerate(TENSOR_PATHS):
        nb = np.linalg.norm(gw_r[i])
        if nb == 0: continue
        errs = [np.linalg.norm(res[k]["nef"][i].cpu().numpy() - gw_r[i]) / nb for k in res]
        print(f"  {'/'.join(path[-3:]):48s}" + "  ".join(f"{e:8.1e}" for e in errs))
    for k in ("p_pos", "a"):
        errs = [np.linalg.norm(res[m]["autodecoder"][k].cpu().numpy() - gl_r[k]) / np.linalg.norm(gl_r[k]) for m in res]
        print(f"  lat0 {k:42s}" + "  ".join(f"{e:8.1e}" for e in errs))
        errs = [np.linalg.norm(res[m]["meta_sgd_lrs"][k].cpu().numpy() - gr_r[k]) / np.linalg.norm(gr_r[k]) for m in res]
        print(f"  lrs  {k:42s}" + "  ".join(f"{e:8.1e}" for e in errs))
