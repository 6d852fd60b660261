import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import enf_ref_np as R, enf_ref_torch as T
from tests.helpers import make_cfg, make_inputs
from tests.test_gpu_layers import _nef
cuda = torch.device("cuda:0")
def run(inv, L, Z, N, needw, freq=(0.5, 1.0), D=64, seed=0):
    cfg = dict(make_cfg(inv, D=D, H=2, C=8, O=2, freq=freq), num_layers=L)
    prm = R.init_params(D + L + seed, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, 2, N, Z, L + seed)
    w = np.random.default_rng(1).standard_normal((2, N, 2))
    rp = T.to_torch(prm, torch.float64)
    rpp = torch.tensor(p, requires_grad=True)
    ref = T.nef_apply(rp, cfg, torch.tensor(x), rpp, torch.tensor(a), torch.tensor(s))
    (ref * torch.tensor(w)).sum().backward()
    nef = _nef(cfg, "f32")
    P = nef.load_params(prm, device=cuda)
    if needw:
        for v in nef.param_tensors(P): v.requires_grad_(True)
    t = lambda v, g=False: torch.tensor(v, dtype=torch.float32, device=cuda, requires_grad=g)
    dpp = t(p, True)
    out = nef.apply(P, t(x), dpp, t(a), t(s))
    (out * t(w)).sum().backward()
    g, r = dpp.grad.cpu().double().numpy(), rpp.grad.numpy()
    e = np.abs(g - r)
    print(inv, "L", L, "Z", Z, "needw", needw, "rel %.2e" % (np.linalg.norm(g - r) / np.linalg.norm(r)), "max abs err %.3e at" % e.max(), np.unravel_index(e.argmax(), e.shape), "ref there %.3f" % r[np.unravel_index(e.argmax(), e.shape)], "|r| max %.1f" % np.abs(r).max())
for seed in range(6):
    run("rel_pos_periodic", 2, 9, 40, False, seed=seed)
for seed in range(3):
    run("rel_pos_periodic", 3, 16, 40, False, seed=seed)
