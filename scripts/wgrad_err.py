"""Per-tensor weight-gradient error of the training path vs the fp64 oracle (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef
from tests.test_gpu_weight_grads import ref, hip
from enf_pde_amd.enf.models import TENSOR_PATHS
cuda = torch.device("cuda:0")
for (D, H, C, O, Z, N, B, seed) in [(128, 2, 16, 1, 64, 256, 3, 192), (64, 2, 16, 1, 16, 100, 3, 80), (128, 2, 16, 1, 64, 1024, 4, 7)]:
    cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=C, O=O)
    prm = R.init_params(seed, cfg, jitter=0.1)
    x, p, a, s = make_inputs(cfg, B, N, Z, seed + 1)
    w = np.random.default_rng(seed + 2).standard_normal((B, N, O))
    ro, rg, rp, ra, rs = ref(prm, cfg, x, p, a, s, w)
    for prec in ("f32", "bf16"):
        ho, hg, hp, ha, hs = hip(cuda, build_nef(cfg, prec), prm, x, p, a, s, w)
        gmax = max(np.linalg.norm(g) for g in rg)
        print(f"--- D{D} H{H} Z{Z} N{N} {prec}")
        for path, g, r in zip(TENSOR_PATHS, hg, rg):
            nr = np.linalg.norm(r)
            if nr == 0: continue
            print(f"  {'/'.join(path[-3:]):60s} |g|/gmax {nr/gmax:9.2e}  relerr {np.linalg.norm(g-r)/nr:9.2e}")
