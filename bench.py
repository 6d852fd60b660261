#!/usr/bin/env python3
"""bench.py -- ENF fit+decode throughput on MI355X (BASELINE.json metric, config 2).

One "step" = the hot path over one meta-batch of synthetic fields resident in HBM:
  fit    : MAML inner loop, S=3 steps of (HIP forward + HIP backward-to-latents) on N_s=512 sampled
           points plus the final forward (reference pde_trainer.py:191-235), for B signals,
  decode : forward on the full 64x64 grid for the B fitted latent sets (pde_trainer.py:397-402).
query points per step = B * ((S+1) * N_s + N).  Weak scaling: every rank runs its own B signals;
the path has no data-path collective, ranks only meet at the timing barrier.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : the dominant kernel (enf_pair_fwd_kernel at the decode shape) timed alone with
                 events on the launch stream; achieved = algorithmic (as-written) per-pair FLOPs of
                 SURVEY.md 8a / 8d per launch / average launch time, against the bf16 MFMA peak.
  cpu_baseline : the un-fused PyTorch-CPU restatement of the reference (oracle/, "port": JAX is
                 not installable here) timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H, C, O, Z = 128, 2, 16, 1, 64          # config_navier_stokes.yaml:39-55 with 64 latents (BASELINE config 2)
GRID = 64
N = GRID * GRID
N_S, S = 512, 3                            # max_num_sampled_points, num_inner_steps (config_navier_stokes.yaml:73,92)
B_PER_GPU = 16                             # batch_size 8 x fit_on_num_steps 2 (config_navier_stokes.yaml:23,80)
PEAK_BF16 = 2.5e15                         # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12


def pair_flops_per_query(z=Z, d=D, h=H, i=4):
    """Per-pair (as-written) part of SURVEY.md 8d's F_query: Z*(10D^2 + 10HD^2 + 2ID + 6HD)."""
    return z * (10 * d * d + 10 * h * d * d + 2 * i * d + 6 * h * d)


def synth_fields(b, seed, device):
    """Band-limited random fields sum_k c_k cos(pi k.x + phi_k), |k_i| <= 4, unit variance."""
    g = torch.Generator().manual_seed(seed)
    lin = torch.linspace(-1, 1, GRID, dtype=torch.float64)
    X, Y = torch.meshgrid(lin, lin, indexing="xy")                   # fit_navier_stokes.py:32-33
    coords = torch.stack([X, Y], -1).reshape(-1, 2)
    ks = torch.stack(torch.meshgrid(torch.arange(-4, 5), torch.arange(-4, 5), indexing="ij"), -1).reshape(-1, 2).double()
    c = torch.randn(b, ks.shape[0], generator=g, dtype=torch.float64)
    ph = torch.rand(b, ks.shape[0], generator=g, dtype=torch.float64) * 2 * math.pi
    f = (c[:, None, :] * torch.cos(math.pi * (coords @ ks.T)[None] + ph[:, None, :])).sum(-1)
    f = f / f.std(dim=1, keepdim=True)
    return coords.float().to(device), f[..., None].float().to(device)


def build(device, precision):
    from types import SimpleNamespace as NS
    from enf_pde_amd.fitting import get_model_pde, default_meta_sgd_lrs, make_masks
    from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta
    cfg = NS(nef=NS(num_in=2, num_out=O, num_layers=0, num_hidden=D, num_heads=H, condition_value_transform=True,
                    latent_dim=C, num_latents=Z, use_gaussian_window=True, embedding_type="rff",
                    embedding_freq_multiplier_invariant=0.05, embedding_freq_multiplier_value=0.1,
                    invariant_type="rel_pos_periodic"))
    nef, _ = get_model_pde(cfg, precision=precision)
    params = nef.init(0, device=device)
    ad = PositionOrientationFeatureAutodecoderMeta(num_signals=1, num_latents=Z, latent_dim=C, num_pos_dims=2,
                                                   num_ori_dims=0, gaussian_window_size=-1, coordinate_system="cartesian")
    lat0 = ad.init(device=device)["params"]
    lrs = default_meta_sgd_lrs(C, 1.0, 5.0, 0.0, device=device)     # config_navier_stokes.yaml:94-96
    masks = make_masks(N, N_S, S, generator=torch.Generator().manual_seed(1), device=device)
    return nef, params, lat0, lrs, masks


def one_step(nef, params, lat0, lrs, coords, img, masks):
    from enf_pde_amd.fitting import inner_loop, decode
    loss, lat = inner_loop(nef, params, lat0, lrs, coords, img, masks)
    recon = decode(nef, params, coords, lat["p_pos"], lat["a"], lat["gaussian_window"])
    return loss, recon


def split_leg(nef, params, lat0, lrs, coords, img, masks, device, iters=10):
    """the fit and the decode halves of a step timed separately (rank 0, outside the timed region):
    qps_fit = B (S+1) N_s / t_fit, qps_decode = B N / t_decode."""
    from enf_pde_amd.fitting import inner_loop, decode
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    _, lat = inner_loop(nef, params, lat0, lrs, coords, img, masks)
    torch.cuda.synchronize(device)
    ev[0].record()
    for _ in range(iters):
        _, lat = inner_loop(nef, params, lat0, lrs, coords, img, masks)
    ev[1].record()
    for _ in range(iters):
        decode(nef, params, coords, lat["p_pos"], lat["a"], lat["gaussian_window"])
    ev[2].record()
    torch.cuda.synchronize(device)
    t_fit, t_dec = ev[0].elapsed_time(ev[1]) / iters * 1e-3, ev[1].elapsed_time(ev[2]) / iters * 1e-3
    return {"qps_fit": round(B_PER_GPU * (S + 1) * N_S / t_fit, 1), "qps_decode": round(B_PER_GPU * N / t_dec, 1),
            "ms_fit": round(t_fit * 1e3, 4), "ms_decode": round(t_dec * 1e3, 4), "n_gpus": 1}


def roofline_leg(nef, params, coords, device, iters=20):
    """Time enf_pair_fwd_kernel alone (decode shape) with events on the launch stream."""
    from enf_pde_amd import _lib
    lib = _lib.load()
    b = B_PER_GPU
    desc = nef._desc(b, N, Z)
    packed = nef.pack(params)
    ws = nef._workspace(desc, device)
    g = torch.Generator().manual_seed(3)
    p = (torch.rand(b, Z, 2, generator=g) * 2 - 1).to(device)
    a = (1 + 0.1 * torch.randn(b, Z, C, generator=g)).to(device)
    sg = torch.full((b, Z, 1), 0.25, device=device)
    out = torch.empty(b, N, O, device=device)
    ybar = torch.empty(b, N, H * D, device=device)
    lse = torch.empty(b, N, H, device=device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(stages):
        _lib.check(lib.enf_forward_stages(ctypes.byref(desc), P(coords), 0, P(p), P(a), P(sg), P(packed), P(out), P(ybar),
                                          P(lse), P(ws), ws.numel(), stages, st))
    run(1 | 8)          # ENF_STAGE_PROLOGUE | ENF_STAGE_FOLD: latent table + per-latent folded matrices stay in the workspace
    for _ in range(3):
        run(2)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(2)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / iters
    nef._ws_touch(ws)           # the workspace no longer holds any autograd graph's latent table
    flops = b * N * pair_flops_per_query()
    bf16 = nef.precision == "bf16"
    peak = PEAK_BF16 if bf16 else PEAK_F32
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pair_fwd_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get("workload") == f"B{b}_N{N}_Z{Z}_{nef.precision}":
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "mfma", "kernel": "enf_pair_fwd_kernel", "achieved": round(flops / (ms * 1e-3) / 1e12, 2),
            "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / peak, 4),
            "traffic": traffic, "launch_ms": round(ms, 4), "flops_per_launch": flops,
            "note": "algorithmic (as-written) per-pair FLOPs; the kernel executes ~0.33x of them (exact folds, DESIGN.md)"
                    + ("" if bf16 else "; in f32 mode the fraction can therefore exceed 1 of the 157 TF fp32-MFMA peak")}


def cpu_baseline_leg(seed=0):
    """PyTorch-CPU un-fused restatement (oracle) on a bounded sample: 4 signals, the 4 fit forwards
    (3 of them with backward-to-latents) on N_s=512 points + decode of the full 4096-point grid."""
    from oracle import enf_ref_np as R
    from oracle import enf_ref_torch as T
    # the GPU box reports 256 logical CPUs but grants a 16-core share; oversubscribing torch's
    # intra-op pool makes the baseline slower, not faster
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    cfg = dict(num_hidden=D, num_heads=H, latent_dim=C, num_out=O, invariant="rel_pos_periodic", num_in=2,
               embedding_freq_multiplier=(0.05, 0.1), use_gaussian_window=True)
    prm = T.to_torch(R.init_params(seed, cfg), torch.float32)
    nb = 4
    lat = {k: torch.tensor(v, dtype=torch.float32) for k, v in R.init_latents(1, Z, C, "rel_pos_periodic").items()}
    coords, img = synth_fields(nb, 5, "cpu")
    masks = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(s))[:N_S] for s in range(S + 1)], 1)
    lrs = {"p_pos": torch.tensor([1.0]), "a": torch.full((C,), 5.0), "gaussian_window": torch.tensor([0.0])}
    n_dec = N
    t0 = time.perf_counter()
    _, fitted = T.inner_loop(prm, cfg, lat, lrs, coords, img, masks)
    with torch.no_grad():
        T.nef_apply_chunked(prm, cfg, coords[None, :n_dec].expand(nb, -1, -1), fitted["p_pos"], fitted["a"], fitted["gaussian_window"], chunk=512)
    dt = time.perf_counter() - t0
    pts = nb * ((S + 1) * N_S + n_dec)
    return {"value": round(pts / dt, 1), "unit": "query-points/s", "cores": cores, "kind": "port",
            "sample": f"{nb} signals: fit (S={S}, N_s={N_S}, fwd+bwd) + decode of {n_dec}/{N} grid points, fp32, "
                      f"chunk 512; PyTorch-CPU restatement of the reference (JAX unavailable); {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from enf_pde_amd.fitting import init_distributed
    rank, world, local_rank = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decoder has no CPU path")
    # one rank per GPU (the driver's launch); ENF_DIST_BACKEND=gloo lets several ranks share one GPU in rehearsals
    device = torch.device("cuda", (local_rank % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(device)

    nef, params, lat0, lrs, masks = build(device, args.precision)
    coords, img = synth_fields(B_PER_GPU, 100 + rank, device)

    for _ in range(args.warmup):
        one_step(nef, params, lat0, lrs, coords, img, masks)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, recon = one_step(nef, params, lat0, lrs, coords, img, masks)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if world > 1:
        on_gpu = torch.distributed.get_backend() == "nccl"
        t = torch.tensor([dt], device=device if on_gpu else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    pts_per_step = world * B_PER_GPU * ((S + 1) * N_S + N)
    result = {
        "metric": "query-points/sec (ENF fit+decode) at 64 latents x 64^2 grid",
        "value": round(pts_per_step * args.steps / dt, 1), "unit": "query-points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision if args.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"navier_stokes_64x64_z{Z}_b{B_PER_GPU}_per_gpu", "grid": [GRID, GRID], "latents": Z,
                   "signals_per_gpu": B_PER_GPU, "inner_steps": S, "sampled_points": N_S, "num_hidden": D,
                   "num_heads": H, "invariant": "rel_pos_periodic", "parallelism": f"dp{world} (signals sharded, no data-path collective)"},
        "final_fit_loss": round(float(loss), 6),
    }
    if rank == 0:
        result["split"] = split_leg(nef, params, lat0, lrs, coords, img, masks, device)
        result["roofline"] = roofline_leg(nef, params, coords, device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_leg()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()          # ranks leave together (rank 0 has just run its extra legs)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
