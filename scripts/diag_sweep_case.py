"""Dev diagnostic: error values of one case of tests/test_gpu_backward.py::test_random_shape_sweep, repeated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tests.test_gpu_backward as T
from enf_pde_amd import _lib
case = int(sys.argv[1]); zf = int(sys.argv[2]); zb = int(sys.argv[3])
lib = _lib.load(); lib.enf_set_zfold(zf); lib.enf_set_zfold_bwd(zb)
cuda = torch.device("cuda:0")
rng = np.random.default_rng(1000 + case)
inv = ["rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos", "rel_pos", "norm_rel_pos", "ball", "ball_lat"][case % 9]
D, H = [(64, 1), (64, 2), (128, 1), (128, 2), (64, 4)][int(rng.integers(5))]
B, N, Z = int(rng.integers(1, 4)), int(rng.integers(1, 150)), int(rng.integers(1, 40))
precision = "f32" if case % 2 == 0 else "bf16"
cfg = T.make_cfg(inv, D=D, H=H, C=int(rng.integers(2, 20)), O=int(rng.integers(1, 5)), freq=(0.3, 0.6))
print(inv, D, H, B, N, Z, precision)
seed = 2000 + case
prm = T.R.init_params(seed, cfg, jitter=0.1)
x, p, a, s = T.make_inputs(cfg, B, N, Z, seed + 1)
w = np.random.default_rng(seed + 2).standard_normal((B, N, cfg["num_out"]))
_, rp, ra, rs = T.ref_grads(prm, cfg, x, p, a, s, w)
for it in range(6):
    if it == 3:
        junk = [torch.full((1 << 22,), float("nan"), device=cuda) for _ in range(8)]; del junk     # poison the allocator's free blocks
    nef = T.build_nef(cfg, precision)
    o, gp, ga, gs = T.hip_grads(cuda, nef, prm, x, p, a, s, w)
    print(it, "p %.4f a %.4f sigma %.4f" % (T.rel(gp, rp), T.rel(ga, ra), T.rel(gs, rs)), flush=True)
