"""Third probe of the family (store_probe.py, fwd_probe.py): the whole TRAINING backward through the model API -- all 46 weight
gradients plus the latent gradients (K2, tail forward / backward, K3 with the activation store, K4, prologue backward, library
GEMMs) -- repeated behind the NaN polluter kernel and compared with the first iteration's to 1e-4 of each tensor's largest value
(sums over atomically accumulated latent gradients differ by ~1e-7; a packed-fp32 event of the kind found in K3 is ~1e-2).
    python scripts/k3_race/train_probe.py [iterations] [D] [H] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef
from enf_pde_amd.fitting.ode_models.ponita_ode_g import kernel_basis

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 500
D, H, prec = (int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (128, 2, "bf16")
cuda = torch.device("cuda:0")
B, N, Z = (int(v) for v in os.environ.get("SHAPE", "4,512,64").split(","))
cfg = make_cfg("rel_pos_periodic", D=D, H=H, C=16, O=1)
prm = R.init_params(1, cfg, jitter=0.1)
x, p, a, s = make_inputs(cfg, B, N, Z, 2)
t = lambda v: torch.tensor(v, dtype=torch.float32, device=cuda)
w = torch.randn(B, N, 1, device=cuda)
nanx = torch.full((65536, 4), float("nan"), device=cuda, requires_grad=True)
K1 = {"kernel": torch.full((340, 128), float("nan"), device=cuda), "bias": torch.full((128,), float("nan"), device=cuda)}
K3 = {"kernel": torch.full((128, 64), float("nan"), device=cuda), "bias": torch.full((64,), float("nan"), device=cuda)}
nef = build_nef(cfg, prec)
P = nef.load_params(prm, device=cuda)
W = nef.param_tensors(P)
for v in W:
    v.requires_grad_(True)
first, bad, worst = None, 0, 0.0
for it in range(n_it):
    if it % 2 == 1:
        kernel_basis(nanx, 3, K1, K3).sum().backward()
    dp, da, ds = t(p).requires_grad_(True), t(a).requires_grad_(True), t(s).requires_grad_(True)
    out = nef.apply(P, t(x), dp, da, ds)
    g = torch.autograd.grad((out * w).sum(), list(W) + [dp, da, ds], allow_unused=True)
    torch.cuda.synchronize()
    g = [torch.zeros(1, device=cuda) if v is None else v for v in g]
    if first is None:
        first = [v.clone() for v in g]
        assert all(torch.isfinite(v).all() for v in g)
        continue
    errs = [float((a_ - b_).abs().max() / b_.abs().max().clamp_min(1e-30)) for a_, b_ in zip(g, first)]
    worst = max(worst, max(errs))
    if max(errs) > 1e-4:
        bad += 1
        k = max(range(len(errs)), key=errs.__getitem__)
        print(f"it {it}: tensor {k} of {len(errs)} off by {errs[k]:.2e}; tensors beyond 1e-4: {[i for i, e in enumerate(errs) if e > 1e-4]}", flush=True)
print(f"training backward <{D},{H},{prec}> B={B} N={N} Z={Z}, {n_it} iterations: {bad} with a gradient tensor beyond 1e-4 of its largest value "
      f"(worst over all {worst:.2e})", flush=True)
