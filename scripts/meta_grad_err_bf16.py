"""bf16 mode: error of the frozen-mask finite-difference meta-gradient vs exact second-order autograd of the oracle, by step."""
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
    nef = build_nef(cfg, "bf16"); params = nef.load_params(prm, device=cuda)
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
    res = {}
    for mode, step in (("none", 0), ("fd", 1e-1), ("fd", 5e-2), ("fd", 2e-2), ("fd", 5e-3)):
        _, g = meta_gradients(nef, params, {k: t(v) for k, v in lat0.items()}, {k: t(v) for k, v in lrs.items()}, t(coords), t(img),
                              torch.tensor(masks, device=cuda), second_order=mode, fd_step=step or 5e-3)
        res[(mode, step)] = g
    print("problem", kw, list(res))
    errs = np.array([[np.linalg.norm(res[k]["nef"][i].cpu().numpy() - gw_r[i]) / max(np.linalg.norm(gw_r[i]), 1e-30) for k in res]
                     for i in range(len(TENSOR_PATHS)) if np.linalg.norm(gw_r[i]) > 0])
    print("  weights: median", "  ".join(f"{e:8.1e}" for e in np.median(errs, 0)), "\n           max   ", "  ".join(f"{e:8.1e}" for e in errs.max(0)))
    for k in ("p_pos", "a"):
        print(f"  lat0 {k:6s}", "  ".join(f"{np.linalg.norm(res[m]['autodecoder'][k].cpu().numpy() - gl_r[k]) / np.linalg.norm(gl_r[k]):8.1e}" for m in res))
        print(f"  lrs  {k:6s}", "  ".join(f"{np.linalg.norm(res[m]['meta_sgd_lrs'][k].cpu().numpy() - gr_r[k]) / np.linalg.norm(gr_r[k]):8.1e}" for m in res))
