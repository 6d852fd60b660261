"""Feasibility: torch.cuda.make_graphed_callables over PonitaODEGen.apply (forward + backward), parity and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
from enf_pde_amd.fitting import get_model_pde
from enf_pde_amd.fitting.trainers.pde_trainer import _leaves, _unflatten
dev = torch.device("cuda:0")
B, Z, C, H, J = 16, 64, 16, 128, 64
cfg = NS(nef=NS(num_in=2, num_out=1, num_layers=0, num_hidden=128, num_heads=2, condition_value_transform=True, latent_dim=C,
                num_latents=Z, use_gaussian_window=True, embedding_type="rff", embedding_freq_multiplier_invariant=0.05,
                embedding_freq_multiplier_value=0.1, invariant_type="rel_pos_periodic"),
         node=NS(name="ponita", num_layers=3, num_hidden=H, widening_factor=2, kernel_size="global", degree=3, basis_dim=J))
_, ode = get_model_pde(cfg)
g = torch.Generator().manual_seed(0)
p = (torch.rand(B, Z, 2, generator=g) * 2 - 1).to(dev).requires_grad_(True)
a = (1 + 0.1 * torch.randn(B, Z, C, generator=g)).to(dev).requires_grad_(True)
w = torch.full((B, Z, 1), 0.25, device=dev)
P = ode.init(0, (p, a, w))
leaves = [t.requires_grad_(True) for t in _leaves(P)]


def fn(p_, a_, *lv):
    dp, da, _ = ode.apply(_unflatten(P, list(lv)), (p_, a_, w))
    return dp, da


def run(f, n):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        dp, da = f(p, a, *leaves)
        gr = torch.autograd.grad((dp ** 2).sum() + (da ** 2).sum(), [p, a] + leaves, allow_unused=True)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3, gr


ms_e, g_e = run(fn, 10)
print("eager ms", ms_e, flush=True)
gf = torch.cuda.make_graphed_callables(fn, (p, a, *leaves), allow_unused_input=True)
print("captured", flush=True)
ms_g, g_g = run(gf, 10)
worst = max(float((x - y).norm() / x.norm().clamp_min(1e-30)) for x, y in zip(g_e, g_g) if x is not None)
print(f"eager {ms_e:.2f} ms  graphed {ms_g:.2f} ms  worst rel grad diff {worst:.2e}")
