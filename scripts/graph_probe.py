import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
from enf_pde_amd.fitting import get_model_pde
dev = torch.device("cuda:0")
B, Z, C, H, J = 16, 64, 16, 128, 64
cfg = NS(nef=NS(num_in=2, num_out=1, num_layers=0, num_hidden=128, num_heads=2, condition_value_transform=True, latent_dim=C,
                num_latents=Z, use_gaussian_window=True, embedding_type="rff", embedding_freq_multiplier_invariant=0.05,
                embedding_freq_multiplier_value=0.1, invariant_type="rel_pos_periodic"),
         node=NS(name="ponita", num_layers=3, num_hidden=H, widening_factor=2, kernel_size="global", degree=3, basis_dim=J))
_, ode = get_model_pde(cfg)
g = torch.Generator().manual_seed(0)
p = (torch.rand(B, Z, 2, generator=g) * 2 - 1).to(dev)
a = (1 + 0.1 * torch.randn(B, Z, C, generator=g)).to(dev)
w = torch.full((B, Z, 1), 0.25, device=dev)
P = ode.init(0, (p, a, w))
with torch.no_grad():
    ref = ode.apply(P, (p, a, w))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            ode.apply(P, (p, a, w))
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = ode.apply(P, (p, a, w))
    gr.replay(); torch.cuda.synchronize()
    print("graph vs eager max diff", float((out[0] - ref[0]).abs().max()), float((out[1] - ref[1]).abs().max()))
    t0 = time.time()
    for _ in range(50): gr.replay()
    torch.cuda.synchronize(); print("graph replay ms", (time.time() - t0) / 50 * 1e3)
    t0 = time.time()
    for _ in range(50): ode.apply(P, (p, a, w))
    torch.cuda.synchronize(); print("eager ms", (time.time() - t0) / 50 * 1e3)
