"""Dev probe (GPU box): error vs oracle and forward timings at the BASELINE config-2 shape."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import enf_ref_np as R
from tests.helpers import make_cfg, make_inputs, build_nef

dev = torch.device("cuda:0")
cfg = make_cfg("rel_pos_periodic", D=128, H=2, C=16, O=1)
prm = R.init_params(0, cfg, jitter=0.1)
t = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
# accuracy on a small slice
x, p, a, s = make_inputs(cfg, 2, 256, 64, 1)
ref = R.nef_apply(prm, cfg, x, p, a, s)
for prec in ("f32", "bf16"):
    nef = build_nef(cfg, prec); params = nef.load_params(prm, device=dev)
    out = nef.apply(params, t(x), t(p), t(a), t(s)).cpu().numpy().astype(np.float64)
    print(prec, "max rel err %.3e  mse %.3e  ref std %.3e" % (np.abs(out-ref).max()/np.abs(ref).max(), ((out-ref)**2).mean(), ref.std()))
# timing
for prec in ("bf16", "f32"):
    nef = build_nef(cfg, prec); params = nef.load_params(prm, device=dev)
    for (B, N) in ((16, 4096), (16, 512), (1, 4096)):
        x, p, a, s = make_inputs(cfg, B, N, 64, 2)
        xt, pt, at, st = t(x), t(p), t(a), t(s)
        for _ in range(3): nef.apply(params, xt, pt, at, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10 if prec == "bf16" else 3
        e0.record()
        for _ in range(iters): nef.apply(params, xt, pt, at, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = B * N * 32.112896e6
        print(f"{prec} B={B} N={N}: {ms:.3f} ms  {B*N/ms*1e3/1e6:.2f} Mq/s  algorithmic {fl/ms/1e9:.1f} TFLOP/s")
