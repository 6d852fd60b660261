"""Dev diagnostic: run the cases of tests/test_gpu_backward.py::test_random_shape_sweep under all four kernel-variant combinations,
interleaved, for several rounds in ONE process and report any result that differs from its first-round value by more than
atomics-order noise (a stale-state / race detector, more sensitive than the oracle tolerance)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tests.test_gpu_backward as T
from enf_pde_amd import _lib
lib = _lib.load()
cuda = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cases = {}
for case in range(10):
    rng = np.random.default_rng(1000 + case)
    inv = ["rel_pos_periodic", "latitude_periodic", "polar_periodic", "ponita", "abs_pos", "rel_pos", "norm_rel_pos", "ball", "ball_lat"][case % 9]
    D, H = [(64, 1), (64, 2), (128, 1), (128, 2), (64, 4)][int(rng.integers(5))]
    if inv in ("ball", "ball_lat"):
        D = 64
    B, N, Z = int(rng.integers(1, 4)), int(rng.integers(1, 150)), int(rng.integers(1, 40))
    precision = "f32" if case % 2 == 0 else "bf16"
    cfg = T.make_cfg(inv, D=D, H=H, C=int(rng.integers(2, 20)), O=int(rng.integers(1, 5)), freq=(0.3, 0.6))
    seed = 2000 + case
    prm = T.R.init_params(seed, cfg, jitter=0.1)
    x, p, a, s = T.make_inputs(cfg, B, N, Z, seed + 1)
    w = np.random.default_rng(seed + 2).standard_normal((B, N, cfg["num_out"]))
    cases[case] = (cfg, precision, prm, x, p, a, s, w)
first, worst, bad = {}, 0.0, 0
order = np.random.default_rng(0)
for rnd in range(rounds):
    keys = [(c, zf, zb) for c in cases for zf in (0, 1) for zb in (0, 1)]
    order.shuffle(keys)
    for c, zf, zb in keys:
        cfg, precision, prm, x, p, a, s, w = cases[c]
        nef = T.build_nef(cfg, precision)
        nef.pair_variants = (("latent_split", "z_fold")[zf], ("latent_split", "z_fold")[zb])
        res = T.hip_grads(cuda, nef, prm, x, p, a, s, w)
        if (c, zf, zb) not in first:
            first[(c, zf, zb)] = res
            continue
        for name, r0, r1 in zip(("out", "dp", "da", "dsigma"), first[(c, zf, zb)], res):
            d = np.linalg.norm(r1 - r0) / max(np.linalg.norm(r0), 1e-30)
            worst = max(worst, d)
            if d > 1e-4:
                bad += 1
                print(f"round {rnd} case {c} zfold {zf} zfold_bwd {zb} {precision} {name}: deviates {d:.3e}", flush=True)
print(f"rounds {rounds}: worst deviation {worst:.3e}, {bad} results beyond 1e-4", flush=True)
