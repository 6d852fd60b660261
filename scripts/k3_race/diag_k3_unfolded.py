"""Dev diagnostic: which ingredient of sweep case 3 (ponita, D=64, H=2, B=1, N=87, Z=11, bf16, unfolded backward) makes its
gradients depend on stale memory.  Each configuration: repeated runs with junk written into freed allocator blocks in between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tests.test_gpu_backward as T
from enf_pde_amd import _lib
lib = _lib.load()
cuda = torch.device("cuda:0")
base = dict(inv="ponita", D=64, H=2, B=1, N=87, Z=11, prec="bf16", C=7, O=2)
variants = [{}, {"inv": "rel_pos"}, {"H": 4}, {"H": 1}, {"D": 128}]
for v in variants:
    c = dict(base, **v)
    cfg = T.make_cfg(c["inv"], D=c["D"], H=c["H"], C=c["C"], O=c["O"], freq=(0.3, 0.6))
    prm = T.R.init_params(5, cfg, jitter=0.1)
    x, p, a, s = T.make_inputs(cfg, c["B"], c["N"], c["Z"], 6)
    w = np.random.default_rng(7).standard_normal((c["B"], c["N"], cfg["num_out"]))
    first, worst = None, {}
    for it in range(25):
        junk = [torch.randn(int(n), device=cuda) * 10 for n in np.random.default_rng(it).integers(1 << 10, 1 << 22, 12)]
        del junk
        nef = T.build_nef(cfg, c["prec"])
        nef.pair_variants = ("latent_split", "latent_split")
        res = T.hip_grads(cuda, nef, prm, x, p, a, s, w)
        if first is None:
            first = res; continue
        for name, r0, r1 in zip(("out", "dp", "da", "dsigma"), first, res):
            worst[name] = max(worst.get(name, 0.0), np.linalg.norm(r1 - r0) / max(np.linalg.norm(r0), 1e-30))
    print(v, {k: f"{e:.1e}" for k, e in worst.items()}, flush=True)
