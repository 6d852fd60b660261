"""Companion of store_probe.py: are the OTHER pair-kernel instantiations affected by the packed-fp32 effect found there?
Through the model API (nef.apply + autograd for the latent gradients), with the NaN polluter kernel in front of every second
iteration: the forward is deterministic (bitwise comparison of the reconstruction), the latent gradients are atomic sums
(compared to 1e-4 of their largest value).   python scripts/k3_race/fwd_probe.py [iterations] [D] [H] [precision] [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef
from enf_pde_amd.fitting.ode_models.ponita_ode_g import kernel_basis

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D, H, prec = (int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (128, 2, "bf16")
variants = sys.argv[5:] or ["latent_split", "z_fold"]
cuda = torch.device("cuda:0")
B, N, Z = 4, 512, 64
cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=16, O=1)
prm = R.init_params(1, cfg, jitter=0.1)
x, p, a, s = make_inputs(cfg, B, N, Z, 2)
t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
w = torch.randn(B, N, 1, device=cuda)
nanx = torch.full((65536, 4), float("nan"), device=cuda, requires_grad=True)
K1 = {"kernel": torch.full((340, 128), float("nan"), device=cuda), "bias": torch.full((128,), float("nan"), device=cuda)}
K3 = {"kernel": torch.full((128, 64), float("nan"), device=cuda), "bias": torch.full((64,), float("nan"), device=cuda)}
for variant in variants:
    nef = build_nef(cfg, prec)
    nef.pair_variants = (variant, variant)
    P = nef.load_params(prm, device=cuda)
    first, bad, worst = None, {"out": 0, "grad": 0}, 0.0
    for it in range(n_it):
        if it % 2 == 1:
            kernel_basis(nanx, 3, K1, K3).sum().backward()
        dp, da, ds = t(p).requires_grad_(True), t(a).requires_grad_(True), t(s).requires_grad_(True)
        out = nef.apply(P, t(x), dp, da, ds)
        (out * w).sum().backward()
        torch.cuda.synchronize()
        g = torch.cat([dp.grad.flatten(), da.grad.flatten(), ds.grad.flatten()])
        if first is None:
            first = (out.detach().clone(), g.clone())
            assert torch.isfinite(out).all() and torch.isfinite(g).all()
            continue
        if not torch.equal(out.detach(), first[0]):
            bad["out"] += 1
            rows = (out.detach() != first[0]).view(B * N, -1).any(1).nonzero().flatten().tolist()
            print(f"{variant} it {it}: output differs in {len(rows)} queries, first {rows[:8]}, max {float((out.detach() - first[0]).abs().max()):.3g}", flush=True)
        e = float((g - first[1]).abs().max() / first[1].abs().max())
        worst = max(worst, e)
        if e > 1e-4:
            bad["grad"] += 1
            print(f"{variant} it {it}: latent gradients off by {e:.2e} of their largest value", flush=True)
    print(f"{variant} <{D},{H},{prec}> B={B} N={N} Z={Z}, {n_it} iterations: forward differs {bad['out']}; latent gradients beyond 1e-4: "
          f"{bad['grad']} (worst {worst:.2e})", flush=True)
