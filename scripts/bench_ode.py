"""Latent ODE at the bench shape (16 signals x 64 latents of width 16; config_navier_stokes.yaml's node: ponita, hidden 128,
basis 64, 3 layers, degree 3): time of one derivative evaluation (forward, forward + backward), of the fused SepGconv
kernels alone against their fp32-MFMA roofline, and of one ode_train_step (10 frames, Euler, 512 points per frame).
Prints one JSON line.  Usage: python scripts/bench_ode.py [iters]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
import bench
from enf_pde_amd.fitting import get_model_pde
from enf_pde_amd.fitting.ode_models import sep_gconv
from enf_pde_amd.fitting.trainers import MetaSGDPDETrainer
from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, Z, C, H, J = bench.B_PER_GPU, bench.Z, bench.C, 128, 64
cfg = NS(nef=NS(num_in=2, num_out=1, num_layers=0, num_hidden=128, num_heads=2, condition_value_transform=True, latent_dim=C,
                num_latents=Z, use_gaussian_window=True, embedding_type="rff", embedding_freq_multiplier_invariant=0.05,
                embedding_freq_multiplier_value=0.1, invariant_type="rel_pos_periodic"),
         node=NS(name="ponita", num_layers=3, num_hidden=H, widening_factor=2, kernel_size="global", degree=3, basis_dim=J))
_, ode = get_model_pde(cfg)
nef, params, lat0, lrs, masks = bench.build(dev, "bf16")
g = torch.Generator().manual_seed(0)
p = (torch.rand(B, Z, 2, generator=g) * 2 - 1).to(dev)
a = (1 + 0.1 * torch.randn(B, Z, C, generator=g)).to(dev)
w = torch.full((B, Z, 1), 0.25, device=dev)
P = ode.init(0, (p, a, w))


def timed(fn, n=iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    ms_fwd = timed(lambda: ode.apply(P, (p, a, w)))
leaves = []


def collect(t):
    for v in t.values():
        collect(v) if isinstance(v, dict) else leaves.append(v.requires_grad_(True))


collect(P)


def fwd_bwd():
    pp, aa = p.clone().requires_grad_(True), a.clone().requires_grad_(True)
    dp, da, _ = ode.apply(P, (pp, aa, w))
    torch.autograd.grad((dp ** 2).sum() + (da ** 2).sum(), leaves + [pp, aa], allow_unused=True)


ms_fb = timed(fwd_bwd)
# the same evaluation as a captured (forward, backward) hipGraph pair: what the trainer's ode / dual steps replay
gp, ga = p.clone().requires_grad_(True), a.clone().requires_grad_(True)
gf = ode.graphed_train(P, (gp, ga, w), 1)[0]


def fwd_bwd_graphed():
    dp, da, _ = gf((gp, ga, w))
    torch.autograd.grad((dp ** 2).sum() + (da ** 2).sum(), leaves + [gp, ga], allow_unused=True)


ms_fb_graph = timed(fwd_bwd_graphed)
if os.environ.get("ODE_PHASE") == "eval":          # profiling aid: only the derivative evaluation
    print(json.dumps({"ms_ode_eval_fwd": ms_fwd, "ms_ode_eval_fwd_bwd": ms_fb, "ms_ode_eval_fwd_bwd_graphed": ms_fb_graph}))
    sys.exit(0)
# the fused convolution alone
A_, KB, W_, b_ = torch.randn(B, Z, H, device=dev), torch.randn(B, Z, Z, J, device=dev), torch.randn(J, H, device=dev), torch.randn(H, device=dev)
with torch.no_grad():
    ms_conv = timed(lambda: sep_gconv(A_, KB, W_, b_), 200)
flops = 2.0 * B * Z * Z * J * H + 2.0 * B * Z * Z * H           # kb @ W, then * a and the sender sum
bytes_ = 4.0 * (B * Z * Z * J + 2 * B * Z * H + J * H)
# one ode_train_step: 10 frames, 512 points per frame
conf = NS(optimizer=NS(learning_rate_enf=1e-4, learning_rate_codes=0.0, learning_rate_ode=1e-3),
          meta=NS(learning_rate_meta_sgd=1e-3, num_inner_steps=3, inner_learning_rate_p=1.0, inner_learning_rate_a=5.0,
                  inner_learning_rate_window=0.0, noise_pos_inner_loop=0.0), nef=NS(optimize_gaussian_window=False),
          training=NS(max_num_sampled_points=512), node=NS(dt=1, method="euler"), dataset=NS(traj_len_train=10, traj_len_out_horizon=4))
coords, img = bench.synth_fields(B, 100, dev)
traj = img.reshape(B, 1, bench.GRID, bench.GRID, bench.O).expand(-1, 14, -1, -1, -1).contiguous()
ad = PositionOrientationFeatureAutodecoderMeta(1, Z, C, 2, 0, gaussian_window_size=-1)
tr = MetaSGDPDETrainer(conf, nef, ad, coords, seed=0, ode_model=ode)
st = tr.init_train_state(params)
state = [st]


def step():
    _, state[0] = tr.ode_train_step(state[0], traj)


ms_step_eager = timed(step, max(3, iters // 4))
if os.environ.get("ODE_PHASE") == "step":          # profiling aid: only the eager train step
    print(json.dumps({"ms_ode_train_step_eager": ms_step_eager}))
    sys.exit(0)
tr.graph_ode_training = True
ms_step = timed(step, max(3, iters // 4))
tr.graph_ode_training = False
with torch.no_grad():
    ms_val = timed(lambda: tr.val_step(state[0], traj), max(3, iters // 4))
print(json.dumps({"workload": f"ponita ODE, B={B} Z={Z} C={C} hidden={H} basis={J} layers=3 degree=3 (340 features)",
                  "ms_ode_eval_fwd": round(ms_fwd, 4), "ms_ode_eval_fwd_bwd": round(ms_fb, 4),
                  "ms_ode_eval_fwd_bwd_graphed": round(ms_fb_graph, 4),
                  "pair_evals_per_s_fwd": round(B * Z * Z / ms_fwd * 1e3, 1),
                  "sep_gconv": {"ms": round(ms_conv, 5), "tflops": round(flops / ms_conv / 1e9, 2), "peak_tflops_fp32_mfma": 157.3,
                                "frac_mfma": round(flops / ms_conv / 1e9 / 157.3, 4), "gbps": round(bytes_ / ms_conv / 1e6, 1),
                                "frac_hbm": round(bytes_ / ms_conv / 1e6 / 8000, 4)},
                  "ms_ode_train_step_10_frames": round(ms_step_eager, 3), "ms_ode_train_step_10_frames_graphed_evals": round(ms_step, 3), "ms_val_step_14_frames_full_grid": round(ms_val, 3)}))
