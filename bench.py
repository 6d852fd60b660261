#!/usr/bin/env python3
"""bench.py -- ENF fit+decode throughput on MI355X (BASELINE.json metric; default workload = config 2).

One "step" = the hot path over one meta-batch of synthetic fields resident in HBM:
  fit    : MAML inner loop, S inner steps of (HIP forward + HIP backward-to-latents) on N_s sampled points plus the
           final forward (reference pde_trainer.py:191-235), for B signals per GPU,
  decode : forward on the full grid for the B fitted latent sets (pde_trainer.py:397-402); config 5 rolls the fitted
           latents out for 40 Euler steps of the latent ODE first and decodes all 41 states on the 256 x 256 grid.
query points per step = B * ((S+1) * N_s + frames * N_decode).  Weak scaling: every rank runs its own B signals; fit and
decode have no data-path collective, ranks only meet at the timing barriers.

`--gpus N` with no torchrun environment re-launches this file under `python -m torch.distributed.run` with N ranks as a
child process, before anything touches the GPU, and exits with its code; with a torchrun environment whose WORLD_SIZE is
not N it exits non-zero.  The line it prints always says how many ranks actually ran (`n_gpus`).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline         : the kernel the step spends most of its time in, timed alone with events on its launch stream:
                     achieved = algorithmic (as-written) FLOPs of SURVEY.md 8a / 8d per launch / average launch time
                     against the dense bf16 MFMA peak; `executed_frac` = the MFMA FLOPs the kernel really issues (exact
                     folds) / time / peak.  `roofline_kernels` holds the same for every pair kernel of the
                     step (K2 at the decode shape, K2 at the fit shape, K3 at the fit shape), `roofline_step` the aggregate.
  meta_step        : one OUTER step of the trainer (meta-gradient through the inner steps, the flat all-reduce of the outer
                     gradients -- inside the timed region --, clip + AdamW): the part of the path that has a collective.
  timing           : `value` is the contract's wall-clock mean over exactly --steps steps; `timing.events` is the same step timed
                     per iteration with events on the launch stream, median of --events-steps (100) steps.
  accuracy         : field MSE of the HIP decode of this run's fitted latents against the fp64 oracle on 2 signals x 256 points.
  cpu_baseline     : the un-fused PyTorch-CPU restatement of the reference (oracle/, "port": JAX is not installable here)
                     timed on this box's host cores on a bounded sample, plus a 1-thread figure.
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15                         # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12

# BASELINE.json configs made concrete (hparams from the reference's experiments/fitting/config_*.yaml).
# B = signals per GPU and step: batch_size x fit_on_num_steps of the yaml (config 4: meta-batch 64 over 8 GPUs).
CONFIGS = {
    1: dict(name="diffusion_32x32_z16", inv="ponita", D=64, H=2, C=16, O=1, Z=16, grid=(32, 32), N_s=1024, S=3, B=32,
            freq=(0.05, 0.01), lr=(1.0, 5.0, 0.0), src="config_diff_plane.yaml:23-96 with 16 latents"),
    2: dict(name="navier_stokes_64x64_z64", inv="rel_pos_periodic", D=128, H=2, C=16, O=1, Z=64, grid=(64, 64), N_s=512, S=3,
            B=16, freq=(0.05, 0.1), lr=(1.0, 5.0, 0.0), src="config_navier_stokes.yaml:23-96 with 64 latents"),
    3: dict(name="shallow_water_sphere_96x48_z128", inv="latitude_periodic", D=128, H=2, C=32, O=3, Z=128, grid=(96, 48),
            N_s=4096, S=3, B=4, freq=(0.05, 0.2), lr=(0.0, 5.0, 0.0), src="config_shallow_water.yaml:23-93 with 128 latents"),
    4: dict(name="navier_stokes_128x128_z128", inv="rel_pos_periodic", D=128, H=2, C=16, O=1, Z=128, grid=(128, 128), N_s=512,
            S=3, B=8, freq=(0.05, 0.1), lr=(1.0, 5.0, 0.0), src="config_navier_stokes.yaml, 128 latents (16 x 8), meta-batch 64 / 8 GPUs"),
    5: dict(name="navier_stokes_64x64_fit_256x256_decode_40_step_rollout", inv="rel_pos_periodic", D=128, H=2, C=16, O=1, Z=64,
            grid=(64, 64), decode_grid=(256, 256), rollout=40, N_s=512, S=3, B=2, freq=(0.05, 0.1), lr=(1.0, 5.0, 0.0),
            src="config_navier_stokes.yaml:23-96 (node: ponita, 3 layers, hidden 128, basis 64, euler dt 1)"),
}
HEADLINE = 2

# ---- config 2 under its round-1 names (scripts/ import these)
_c2 = CONFIGS[HEADLINE]
D, H, C, O, Z = _c2["D"], _c2["H"], _c2["C"], _c2["O"], _c2["Z"]
GRID = _c2["grid"][0]
N = GRID * GRID
N_S, S = _c2["N_s"], _c2["S"]
B_PER_GPU = _c2["B"]


def invariant_dim(inv):
    return {"rel_pos_periodic": 4, "latitude_periodic": 4, "polar_periodic": 1, "ponita": 2}[inv]


def pair_flops(c):
    """As-written FLOPs of one (query, latent) pair, forward: 10D^2 + 10HD^2 + 2ID + 6HD."""
    d, h, i = c["D"], c["H"], invariant_dim(c["inv"])
    return 10 * d * d + 10 * h * d * d + 2 * i * d + 6 * h * d


def query_flops(c):
    """Per-query (tail) part of SURVEY.md 8d's F_query: 6(HD)^2 + 2HD.D + 2D^2 + 2DO."""
    d, hd, o = c["D"], c["H"] * c["D"], c["O"]
    return 6 * hd * hd + 2 * hd * d + 2 * d * d + 2 * d * o


def pair_flops_per_query(z=Z, d=D, h=H, i=4):
    return z * (10 * d * d + 10 * h * d * d + 2 * i * d + 6 * h * d)


# D x D MFMA GEMMs a pair kernel issues per pair (counted from the kernels at H = 2): executed FLOPs = units * 2 D^2
MFMA_UNITS = {("fwd", "z_fold"): lambda h: 3 + h, ("fwd", "z_fold_zsplit"): lambda h: 3 + h, ("fwd", "latent_split"): lambda h: 3 + 3 * h,
              ("bwd", "z_fold"): lambda h: 16 if h == 2 else None, ("bwd", "latent_split"): lambda h: 24 if h == 2 else None}


def coords_of(c, grid, device):
    """Query grid of a config: Cartesian 'xy' meshgrid on [-1, 1]^2 (fit_navier_stokes.py:32-33) or the (phi, theta) sphere
    grid, 'ij', phi in [0, 2 pi), theta in (0, pi) (datasets/pdes.py:587-589)."""
    w, h = grid
    if c["inv"] in ("latitude_periodic", "polar_periodic"):
        phi = torch.arange(w, dtype=torch.float64) * (2 * math.pi / w)
        th = (torch.arange(h, dtype=torch.float64) + 0.5) * (math.pi / h)
        P, T = torch.meshgrid(phi, th, indexing="ij")
        return torch.stack([P, T], -1).reshape(-1, 2).float().to(device)
    X, Y = torch.meshgrid(torch.linspace(-1, 1, w, dtype=torch.float64), torch.linspace(-1, 1, h, dtype=torch.float64), indexing="xy")
    return torch.stack([X, Y], -1).reshape(-1, 2).float().to(device)


def synth_targets(c, coords, b, seed, device):
    """Band-limited random fields sum_k c_k cos(k.x' + phi_k), |k_i| <= 4, unit variance per channel;
    x' = pi x on the plane, (phi, 2 theta) on the sphere (periodic in both)."""
    g = torch.Generator().manual_seed(seed)
    x = coords.double().cpu()
    x = torch.stack([x[:, 0], 2 * x[:, 1]], -1) if c["inv"] in ("latitude_periodic", "polar_periodic") else math.pi * x
    ks = torch.stack(torch.meshgrid(torch.arange(-4, 5), torch.arange(-4, 5), indexing="ij"), -1).reshape(-1, 2).double()
    ph = x @ ks.T                                                           # (N, K)
    out = []
    for _ in range(c["O"]):
        cf = torch.randn(b, ks.shape[0], generator=g, dtype=torch.float64)
        off = torch.rand(b, ks.shape[0], generator=g, dtype=torch.float64) * 2 * math.pi
        f = (cf[:, None, :] * torch.cos(ph[None] + off[:, None, :])).sum(-1)
        out.append(f / f.std(dim=1, keepdim=True))
    return torch.stack(out, -1).float().to(device)


def synth_fields(b, seed, device):
    """Config 2's grid and targets (round-1 signature)."""
    coords = coords_of(_c2, _c2["grid"], device)
    return coords, synth_targets(_c2, coords, b, seed, device)


def _Autodecoder(c):
    """The meta autodecoder of a config; config 4's 128 latents sit on a 16 x 8 cell-centred grid (the reference's
    init_positions_grid asserts a square number), window = d / sqrt(Z) by the same d / k rule (autodecoder.py:38-43)."""
    from enf_pde_amd.enf.latents.autodecoder_meta import PositionOrientationFeatureAutodecoderMeta as AD
    n_ori = 1 if c["inv"] == "ponita" else 0
    system = "polar" if c["inv"] in ("latitude_periodic", "polar_periodic") else "cartesian"
    ad = AD(num_signals=1, num_latents=c["Z"], latent_dim=c["C"], num_pos_dims=2, num_ori_dims=n_ori,
            gaussian_window_size=-1, coordinate_system=system)
    k = round(c["Z"] ** 0.5)
    if system == "cartesian" and k * k != c["Z"]:
        def init(key=None, device="cuda", _z=c["Z"], _C=c["C"]):
            nx = 2 ** math.ceil(math.log2(_z) / 2)
            ny = _z // nx
            ax = lambda n: torch.linspace(-1 + 1 / n, 1 - 1 / n, n, dtype=torch.float64)
            g = torch.stack(torch.meshgrid(ax(nx), ax(ny), indexing="ij"), -1).reshape(1, -1, 2).float()
            return {"params": {"p_pos": g.to(device), "a": torch.ones(1, _z, _C, device=device),
                               "gaussian_window": torch.full((1, _z, 1), 2.0 / math.sqrt(_z), device=device)}}
        ad.init = init
    return ad


def model_config(c):
    return NS(nef=NS(num_in=2, num_out=c["O"], num_layers=0, num_hidden=c["D"], num_heads=c["H"], condition_value_transform=True,
                     latent_dim=c["C"], num_latents=c["Z"], use_gaussian_window=True, embedding_type="rff",
                     embedding_freq_multiplier_invariant=c["freq"][0], embedding_freq_multiplier_value=c["freq"][1],
                     invariant_type=c["inv"], optimize_gaussian_window=False),
              node=(NS(name="ponita", num_layers=3, num_hidden=128, widening_factor=2, degree=3, basis_dim=64) if c.get("rollout") else None),
              optimizer=NS(learning_rate_enf=1e-4, learning_rate_codes=0.0),
              meta=NS(learning_rate_meta_sgd=1e-4, num_inner_steps=c["S"], inner_learning_rate_p=c["lr"][0],
                      inner_learning_rate_a=c["lr"][1], inner_learning_rate_window=c["lr"][2], noise_pos_inner_loop=0.0),
              training=NS(max_num_sampled_points=c["N_s"]))


def build_config(c, device, precision):
    """Model, random-init weights, meta-init latents, inner learning rates, sampling masks (and the ODE model of config 5)."""
    from enf_pde_amd.fitting import get_model_pde, default_meta_sgd_lrs, make_masks
    cfg = model_config(c)
    nef, ode = get_model_pde(cfg, precision=precision)
    params = nef.init(0, device=device)
    ad = _Autodecoder(c)
    lat0 = ad.init(device=device)["params"]
    lrs = default_meta_sgd_lrs(c["C"], *c["lr"], with_ori="p_ori" in lat0, device=device)
    n = c["grid"][0] * c["grid"][1]
    masks = make_masks(n, min(c["N_s"], n), c["S"], generator=torch.Generator().manual_seed(1), device=device)
    ode_params = None
    if ode is not None:
        p0 = torch.cat((lat0["p_pos"], lat0["p_ori"]), -1) if "p_ori" in lat0 else lat0["p_pos"]
        ode_params = ode.init(1, (p0, lat0["a"], lat0["gaussian_window"]), device=device)
    return NS(cfg=cfg, nef=nef, params=params, ad=ad, lat0=lat0, lrs=lrs, masks=masks, ode=ode, ode_params=ode_params)


def build(device, precision):
    """Config 2 (round-1 signature)."""
    m = build_config(_c2, device, precision)
    return m.nef, m.params, m.lat0, m.lrs, m.masks


def one_step(nef, params, lat0, lrs, coords, img, masks):
    from enf_pde_amd.fitting import inner_loop, decode
    loss, lat = inner_loop(nef, params, lat0, lrs, coords, img, masks)
    recon = decode(nef, params, coords, lat["p_pos"], lat["a"], lat["gaussian_window"])
    return loss, recon


def pose_of(lat):
    return torch.cat((lat["p_pos"], lat["p_ori"]), -1) if "p_ori" in lat else lat["p_pos"]


def fit(m, coords, img):
    from enf_pde_amd.fitting import inner_loop
    return inner_loop(m.nef, m.params, m.lat0, m.lrs, coords, img, m.masks)


def decode_step(c, m, dcoords, lat):
    """Decode the fitted latents on the decode grid; config 5: 40 Euler steps of the latent ODE first, all 41 states decoded."""
    from enf_pde_amd.fitting import decode
    p, a, s = pose_of(lat), lat["a"], lat["gaussian_window"]
    if c.get("rollout"):
        from enf_pde_amd.fitting import solve_latent_ode
        with torch.no_grad():
            # the inference roll-out of the trainer (MetaSGDPDETrainer.rollout(graph=True), what val_step runs): every
            # derivative evaluation replays ONE captured hipGraph (PonitaODEGen.graphed), captured once per shape
            if getattr(m, "ode_graph", None) is None:
                m.ode_graph = m.ode.graphed(m.ode_params, (p, a, s))
            f = lambda z, _t: m.ode_graph(z)
            tp, ta, ts = solve_latent_ode(f, (p, a, s), 0, c["rollout"], 1, method="euler")     # (B, 41, Z, .)
        p, a, s = (v.reshape(-1, *v.shape[2:]).contiguous() for v in (tp, ta, ts))
    return decode(m.nef, m.params, dcoords, p, a, s)


def step(c, m, coords, dcoords, img):
    loss, lat = fit(m, coords, img)
    return loss, decode_step(c, m, dcoords, lat)


def points_per_signal(c):
    nd = (c.get("decode_grid") or c["grid"])
    frames = c.get("rollout", 0) + 1
    n = c["grid"][0] * c["grid"][1]
    return (c["S"] + 1) * min(c["N_s"], n) + frames * nd[0] * nd[1]


def split_leg(c, m, coords, dcoords, img, device, iters=10):
    """the fit and the decode halves of a step timed separately (rank 0, outside the timed region):
    qps_fit = B (S+1) N_s / t_fit, qps_decode = B frames N / t_decode."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    _, lat = fit(m, coords, img)
    decode_step(c, m, dcoords, lat)
    torch.cuda.synchronize(device)
    ev[0].record()
    for _ in range(iters):
        _, lat = fit(m, coords, img)
    ev[1].record()
    for _ in range(iters):
        decode_step(c, m, dcoords, lat)
    ev[2].record()
    torch.cuda.synchronize(device)
    t_fit, t_dec = ev[0].elapsed_time(ev[1]) / iters * 1e-3, ev[1].elapsed_time(ev[2]) / iters * 1e-3
    n = c["grid"][0] * c["grid"][1]
    fit_pts = c["B"] * (c["S"] + 1) * min(c["N_s"], n)
    return {"qps_fit": round(fit_pts / t_fit, 1), "qps_decode": round((c["B"] * points_per_signal(c) - fit_pts) / t_dec, 1),
            "ms_fit": round(t_fit * 1e3, 4), "ms_decode": round(t_dec * 1e3, 4), "n_gpus": 1}


# ---------------------------------------------------------------------------------------------------------- accuracy / timing
def accuracy_leg(c, m, coords, lat, device, signals=2, queries=256):
    """Field MSE of the HIP decode against the fp64 oracle on a small sub-sample of THIS run's fitted latents (BASELINE.json:
    'field MSE vs JAX ref'; the oracle is the CPU restatement of the reference, parity unpinned -- DESIGN.md 2), outside the
    timed region, rank 0.  The oracle is the checker here, as in the cpu_baseline leg; nothing of it is timed or shipped."""
    import numpy as np
    from oracle import enf_ref_np as R
    cfg = dict(num_hidden=c["D"], num_heads=c["H"], latent_dim=c["C"], num_out=c["O"], invariant=c["inv"], num_in=2,
               embedding_freq_multiplier=c["freq"], use_gaussian_window=True)
    nb = min(signals, c["B"])
    idx = torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(11))[:queries].to(device)
    x = coords[idx][None].expand(nb, -1, -1).contiguous()
    p, a, s = (v[:nb].contiguous() for v in (pose_of(lat), lat["a"], lat["gaussian_window"]))
    with torch.no_grad():
        got = m.nef.apply(m.params, x, p, a, s).double().cpu().numpy()

    def to_np(t):
        return {k: to_np(v) if isinstance(v, dict) else v.detach().double().cpu().numpy() for k, v in t.items()}
    ref = R.nef_apply(to_np(m.params), cfg, *(v.double().cpu().numpy() for v in (x, p, a, s)))
    err = got - ref
    return {"mse_vs_oracle": float((err ** 2).mean()), "max_abs_err": float(np.abs(err).max()), "field_rms": float(np.sqrt((ref ** 2).mean())),
            "n": int(err.size), "sample": f"{nb} fitted signals x {queries} grid points, {m.nef.precision} kernels vs fp64 oracle",
            "budget": 1e-5}


def events_leg(c, m, coords, dcoords, img, device, steps=100):
    """The step timed per iteration with events on the launch stream (torch's current stream: the library joins its side
    stream back into it before returning), `steps` iterations after the timed region: median and mean.  The contract's
    `value` stays the wall-clock figure over exactly --steps steps between barriers; this is the BASELINE.md 2.2 protocol
    (events, median of >= 100) beside it."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize(device)
    ev[0].record()
    for i in range(steps):
        step(c, m, coords, dcoords, img)
        ev[i + 1].record()
    torch.cuda.synchronize(device)
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    med = ts[steps // 2] if steps % 2 else 0.5 * (ts[steps // 2 - 1] + ts[steps // 2])
    pts = c["B"] * points_per_signal(c)
    return {"steps": steps, "ms_median": round(med, 4), "ms_mean": round(sum(ts) / steps, 4), "ms_min": round(ts[0], 4),
            "ms_p90": round(ts[int(0.9 * (steps - 1))], 4), "qps_median_per_gpu": round(pts / (med * 1e-3), 1),
            "clock": "hipEvent pairs around each step on the launch stream, rank 0, after the timed region"}


def graph_leg(c, m, coords, dcoords, img, device, steps=100):
    """--graph-leg (off by default): the whole fit + decode step captured ONCE as a hipGraph (torch.cuda.graph: the library's launches go to
    torch's current stream, its side stream is forked from and joined back into it inside each call, so the capture sees every kernel)
    and replayed `steps` times: what the step costs without the host's launch path and with the graph's tighter kernel spacing.
    Not used for `value`."""
    out = {}
    try:
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                step(c, m, coords, dcoords, img)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, recon = step(c, m, coords, dcoords, img)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize(device)
        l_eager, r_eager = step(c, m, coords, dcoords, img)
        torch.cuda.synchronize(device)
        g.replay()
        torch.cuda.synchronize(device)
        out["max_abs_diff_vs_eager"] = float((recon - r_eager).abs().max())
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            g.replay()
            ev[i + 1].record()
        torch.cuda.synchronize(device)
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize(device)
        wall = (time.perf_counter() - t0) / steps * 1e3
        out.update({"steps": steps, "ms_median": round(ts[steps // 2], 4), "ms_min": round(ts[0], 4), "ms_wall": round(wall, 4),
                    "qps_wall_per_gpu": round(c["B"] * points_per_signal(c) / (wall * 1e-3), 1), "launch": "hipGraph replay of one captured step"})
    except Exception as e:        # a capture the runtime refuses must not cost the bench line
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


# ---------------------------------------------------------------------------------------------------------- roofline
def _time_launches(fn, device, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)     # on torch's current stream,
    e0.record()                                                                             # which is the launch stream
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / iters


def _traffic_profile(kernel, workload):
    """HBM bytes per launch from the committed PMC profile of this kernel and workload, if there is one (the counters need
    rocprofv3 --pmc passes of their own, MI355X_MICROARCH.md; nothing is measured in-run, so `traffic` stays null)."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not name.endswith("_traffic.json"):
            continue
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if t.get("workload") == workload and kernel in t.get("kernel", ""):
            return {"file": "profiles/" + name, "hbm_bytes_per_launch": t.get("hbm_bytes_per_launch")}
    return None


def pair_kernel_rooflines(c, m, device, iters=20):
    """Every pair kernel of a step timed alone (events on the launch stream) at the shapes the step runs it at."""
    from enf_pde_amd import _lib
    lib, nef = _lib.load(), m.nef
    b, z = c["B"], c["Z"]
    frames = c.get("rollout", 0) + 1
    nd = c.get("decode_grid") or c["grid"]
    n_fit, n_dec = min(c["N_s"], c["grid"][0] * c["grid"][1]), nd[0] * nd[1]
    packed = nef.pack(m.params)
    bf16 = nef.precision == "bf16"
    peak = PEAK_BF16 if bf16 else PEAK_F32
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    g = torch.Generator().manual_seed(3)
    out = []

    def shape_leg(tag, bb, n, calls_per_step, with_backward):
        desc = nef._desc(bb, n, z)
        lat = {k: v.repeat_interleave(bb, 0) for k, v in m.lat0.items()}
        p = (pose_of(lat) + 0.02 * torch.randn(pose_of(lat).shape, generator=g).to(device)).contiguous()
        a = (1 + 0.1 * torch.randn(bb, z, c["C"], generator=g)).to(device)
        sg = lat["gaussian_window"].contiguous()
        x = coords_of(c, nd if tag == "decode" else c["grid"], device)[:n].contiguous()
        HD = c["H"] * c["D"]
        o = torch.empty(bb, n, c["O"], device=device)
        ybar, lse = torch.empty(bb, n, HD, device=device), torch.empty(bb, n, c["H"], device=device)
        ws = torch.empty(int(lib.enf_workspace_bytes(ctypes.byref(desc))), device=device, dtype=torch.uint8)
        fwd = lambda stages: _lib.check(lib.enf_forward_stages(ctypes.byref(desc), P(x), 0, P(p), P(a), P(sg), P(packed), P(o),
                                                               P(ybar), P(lse), P(ws), ws.numel(), stages, st))
        fwd(1 | 8 | 2 | 4 | 16)      # prologue, fold, pair, tail (+ stash): the workspace now holds what PAIR alone reuses
        ms = _time_launches(lambda: fwd(2), device, iters)
        variant = {1: "latent_split", 2: "z_fold", 3: "z_fold_zsplit"}[lib.enf_pair_variant(ctypes.byref(desc), 0)]
        pairs = bb * n * z
        rec = lambda kernel, kind, var, ms_, mult, calls: {
            "kernel": kernel, "shape": {"signals": bb, "queries": n, "latents": z}, "variant": var, "launch_ms": round(ms_, 4),
            "launches_per_step": calls, "bound": "mfma", "unit": "TFLOP/s", "peak": peak / 1e12,
            "flops_per_launch": pairs * pair_flops(c) * mult,
            "achieved": round(pairs * pair_flops(c) * mult / (ms_ * 1e-3) / 1e12, 2),
            "frac": round(pairs * pair_flops(c) * mult / (ms_ * 1e-3) / peak, 4),
            "executed_flops_per_launch": (None if MFMA_UNITS[(kind, var)](c["H"]) is None else
                                          pairs * MFMA_UNITS[(kind, var)](c["H"]) * 2 * c["D"] ** 2),
            "traffic": None,
            "traffic_profile": _traffic_profile(kernel, f"B{bb}_N{n}_Z{z}_{nef.precision}")}
        r = rec("enf_pair_fwd_kernel", "fwd", variant, ms, 1, calls_per_step)
        out.append((tag + "_fwd", r))
        if with_backward:
            dout = torch.randn(bb, n, c["O"], generator=g).to(device) / (bb * n)
            dp, da, ds = torch.empty_like(p), torch.empty_like(a), torch.empty_like(sg)
            bwd = lambda flags: _lib.check(lib.enf_backward_latents_ex(ctypes.byref(desc), P(x), 0, P(p), P(a), P(sg), P(packed),
                                                                       P(ybar), P(lse), P(dout), P(dp), P(da), P(ds), P(ws),
                                                                       ws.numel(), flags, st))
            bwd(1 | 2)                # a complete backward on the forward's workspace leaves what ENF_BWD_ONLY_PAIR replays
            ms3 = _time_launches(lambda: bwd(1 | 2 | 8), device, iters)
            var3 = {1: "latent_split", 2: "z_fold"}[lib.enf_pair_variant(ctypes.byref(desc), 1)]
            out.append((tag + "_bwd", rec("enf_pair_bwd_kernel", "bwd", var3, ms3, 2, c["S"])))
        nef._ws_touch(ws)

    shape_leg("decode", b * frames, n_dec, 1, False)
    shape_leg("fit", b, n_fit, c["S"] + 1, True)
    for _, r in out:
        ex = r["executed_flops_per_launch"]
        r["executed_frac"] = None if ex is None else round(ex / (r["launch_ms"] * 1e-3) / peak, 4)
    return out


def step_roofline(c, ms_per_step, precision):
    """The whole step against the MFMA peak: as-written FLOPs of SURVEY.md 8d (F_query incl. the tail; backward = 2 x forward)."""
    n_fit = min(c["N_s"], c["grid"][0] * c["grid"][1])
    nd = c.get("decode_grid") or c["grid"]
    fq = c["Z"] * pair_flops(c) + query_flops(c)
    flops = c["B"] * fq * (n_fit * ((c["S"] + 1) + 2 * c["S"]) + (c.get("rollout", 0) + 1) * nd[0] * nd[1])
    peak = PEAK_BF16 if precision == "bf16" else PEAK_F32
    return {"bound": "mfma", "flops_per_step": flops, "achieved": round(flops / (ms_per_step * 1e-3) / 1e12, 2),
            "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(flops / (ms_per_step * 1e-3) / peak, 4),
            "note": "as-written FLOPs of one step (S+1 forwards and S backwards (= 2 x forward) on the sampled points + the decode) "
                    "/ measured step time, per GPU"}


def roofline_leg(nef, params, coords, device, iters=20):
    """K2 at config 2's decode shape (round-1 signature; scripts/ use it)."""
    m = build_config(_c2, device, nef.precision)
    m.nef, m.params = nef, params
    return dict(pair_kernel_rooflines(_c2, m, device, iters))["decode_fwd"]


# ---------------------------------------------------------------------------------------------------------- outer step
def meta_leg(c, m, coords, img, device, world, steps=5, warmup=2):
    """One outer step of MetaSGDPDETrainer.nef_train_step per iteration: meta-gradient (adjoint recursion through the S inner
    steps, finite-difference second-order terms), ONE flat all-reduce of the outer gradients over the ranks, clip + AdamW /
    Adam updates -- all inside the timed region."""
    from enf_pde_amd.fitting import MetaSGDPDETrainer
    tr = MetaSGDPDETrainer(m.cfg, m.nef, m.ad, coords, seed=0)
    state = tr.init_train_state(nef_params=m.params)
    batch = img.reshape(img.shape[0], *c["grid"][::-1], c["O"])
    for _ in range(warmup):
        loss, state = tr.nef_train_step(state, batch)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, state = tr.nef_train_step(state, batch)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    dt = _max_over_ranks(dt, device, world)
    nbytes = 4 * (sum(t.numel() for t in m.nef.param_tensors(m.params)) + sum(v.numel() for v in m.lat0.values()) + 64)
    return {"ms_per_step": round(dt / steps * 1e3, 3), "signals_per_s": round(world * c["B"] * steps / dt, 1), "steps": steps,
            "n_gpus": world, "loss": round(float(loss), 6), "second_order": tr.second_order,
            "collective": (f"one flat all-reduce of {nbytes / 1e6:.2f} MB per step inside the timed region ({torch.distributed.get_backend()})"
                           if world > 1 else "none at 1 rank (the all-reduce is a no-op)")}


# ------------------------------------------------------------------------------------------------- latent ODE (8f-2)
def ode_leg(c, device, iters=20):
    """One derivative evaluation of the latent ODE of the config's yaml (PonitaODEGen: hidden 128, basis 64, 3 layers, degree 3) at
    the bench's latent shape (B signals x Z latents): forward (inference roll-out form: one replayed hipGraph), forward + backward
    eager and as a captured (forward, backward) graph pair -- SURVEY.md 8f-2, next to the hot path's line (scripts/bench_ode.py has
    the trainer steps).  Rank 0, one GPU."""
    from enf_pde_amd.fitting import get_model_pde
    cfg = model_config(dict(c, rollout=1))
    _, ode = get_model_pde(cfg)
    B, Z, C = c["B"], c["Z"], c["C"]
    g = torch.Generator().manual_seed(0)
    p = (torch.rand(B, Z, 2, generator=g) * 2 - 1).to(device)
    a = (1 + 0.1 * torch.randn(B, Z, C, generator=g)).to(device)
    w = torch.full((B, Z, 1), 0.25, device=device)
    P = ode.init(0, (p, a, w), device=device)
    leaves = []

    def collect(t):
        for v in t.values():
            collect(v) if isinstance(v, dict) else leaves.append(v.requires_grad_(True))
    collect(P)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1) / iters
    with torch.no_grad():
        f = ode.graphed(P, (p, a, w))
        ms_fwd = timed(lambda: f((p, a, w)))
    gp, ga = p.clone().requires_grad_(True), a.clone().requires_grad_(True)

    def fb(fn):
        dp, da, _ = fn((gp, ga, w))
        torch.autograd.grad((dp ** 2).sum() + (da ** 2).sum(), leaves + [gp, ga], allow_unused=True)
    ms_fb = timed(lambda: fb(lambda z: ode.apply(P, z)))
    gt = ode.graphed_train(P, (gp, ga, w), 1)[0]
    ms_fbg = timed(lambda: fb(gt))
    return {"workload": f"PonitaODEGen derivative evaluation, {B} signals x {Z} latents ({B * Z * Z} latent pairs), hidden 128, basis 64, 3 layers",
            "ms_forward_graphed": round(ms_fwd, 4), "ms_forward_backward_eager": round(ms_fb, 4),
            "ms_forward_backward_graphed": round(ms_fbg, 4), "latent_pairs_per_s_forward": round(B * Z * Z / ms_fwd * 1e3, 1),
            "dtype": "f32", "note": "fp32 MFMA kernels (csrc/enf_ode*.hip); eager is host-bound, see DESIGN.md 5b"}


def _max_over_ranks(dt, device, world):
    if world == 1:
        return dt
    on_gpu = torch.distributed.get_backend() == "nccl"
    t = torch.tensor([dt], device=device if on_gpu else "cpu", dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_sample(c, nb, n_dec, seed=0):
    """nb signals through the oracle: the S+1 fit forwards (S of them with backward-to-latents) on N_s points + decode of
    the first n_dec grid points, fp32, decode chunked at 512 queries like pde_trainer.py:397."""
    from oracle import enf_ref_np as R
    from oracle import enf_ref_torch as T
    cfg = dict(num_hidden=c["D"], num_heads=c["H"], latent_dim=c["C"], num_out=c["O"], invariant=c["inv"], num_in=2,
               embedding_freq_multiplier=c["freq"], use_gaussian_window=True)
    prm = T.to_torch(R.init_params(seed, cfg), torch.float32)
    lat = {k: v.float() for k, v in _Autodecoder(c).init(device="cpu")["params"].items()}      # the meta-init (input data)
    coords = coords_of(c, c["grid"], "cpu")
    img = synth_targets(c, coords, nb, 5, "cpu")
    n = coords.shape[0]
    n_s = min(c["N_s"], n)
    masks = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(s))[:n_s] for s in range(c["S"] + 1)], 1)
    lrs = {"p_pos": torch.tensor([c["lr"][0]]), "a": torch.full((c["C"],), c["lr"][1]), "gaussian_window": torch.tensor([c["lr"][2]])}
    if "p_ori" in lat:
        lrs["p_ori"] = torch.tensor([c["lr"][0]])
    t0 = time.perf_counter()
    _, fitted = T.inner_loop(prm, cfg, lat, lrs, coords, img, masks)
    pose = torch.cat((fitted["p_pos"], fitted["p_ori"]), -1) if "p_ori" in fitted else fitted["p_pos"]
    with torch.no_grad():
        T.nef_apply_chunked(prm, cfg, coords[None, :n_dec].expand(nb, -1, -1), pose, fitted["a"], fitted["gaussian_window"], chunk=512)
    dt = time.perf_counter() - t0
    return nb * ((c["S"] + 1) * n_s + n_dec), dt


def cpu_baseline_leg(c=None):
    """PyTorch-CPU un-fused restatement (oracle) on a bounded sample, on all granted cores and on one thread."""
    c = c or _c2
    # the GPU box reports 256 logical CPUs but grants a 16-core share; oversubscribing torch's intra-op pool makes the
    # baseline slower, not faster
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    n = c["grid"][0] * c["grid"][1]
    work = c["Z"] * pair_flops(c) / (64 * 494080)              # sample sizes scaled to config 2's ~19 s / 4 signals
    nb = max(1, min(4, round(4 / work)))
    torch.set_num_threads(cores)
    pts, dt = _cpu_sample(c, nb, n)
    torch.set_num_threads(1)
    n1 = min(n, 1024)
    pts1, dt1 = _cpu_sample(c, 1, n1)
    torch.set_num_threads(cores)
    return {"value": round(pts / dt, 1), "unit": "query-points/s", "cores": cores, "kind": "port",
            "sample": f"{nb} signals: fit (S={c['S']}, N_s={min(c['N_s'], n)}, fwd+bwd) + decode of {n}/{n} grid points, fp32, "
                      f"chunk 512; PyTorch-CPU restatement of the reference (JAX unavailable); {dt:.1f} s",
            "one_thread": {"value": round(pts1 / dt1, 1), "cores": 1,
                           "sample": f"1 signal: the same fit + decode of {n1}/{n} grid points; {dt1:.1f} s"}}


# ---------------------------------------------------------------------------------------------------------- launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def ensure_world(args):
    """Make the number of ranks equal to --gpus, or fail: nothing here touches the GPU (torch.cuda.device_count() does not
    initialise it on this image), so re-launching under torch.distributed.run as a child process is allowed."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}; refusing to report a "
                             f"{args.gpus}-GPU number from {env_world} rank(s)\n")
            sys.exit(2)
        return
    if args.gpus == 1:
        return
    share = os.environ.get("ENF_BENCH_SHARE_GPU") == "1"       # rehearsal: several ranks on one GPU (gloo), never a result
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not share:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node has {ndev} GPU(s)\n")
        sys.exit(2)
    env = dict(os.environ)
    if share:
        env["ENF_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=HEADLINE, choices=sorted(CONFIGS), help="BASELINE.json config (default: 2, the metric's)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-meta", action="store_true", help="skip the outer-step leg")
    ap.add_argument("--no-ode", action="store_true", help="skip the latent-ODE evaluation leg (config 2, one GPU)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel legs")
    ap.add_argument("--kernel-iters", type=int, default=20, help="launches per timed per-kernel leg")
    ap.add_argument("--events-steps", type=int, default=100, help="steps of the per-step event timing (median) reported beside `value`; 0 = skip")
    ap.add_argument("--graph-leg", action="store_true", help="also time the step as a captured hipGraph (reported under timing.graph; `value` unchanged)")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the field-MSE-vs-oracle check of the fitted latents")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run ONLY the per-kernel legs and print them (for rocprofv3 --kernel-trace --stats: the profile then "
                         "holds exactly the launches the legs time, so its per-kernel averages are the legs' launch_ms)")
    args = ap.parse_args()
    ensure_world(args)

    from enf_pde_amd.fitting import init_distributed
    rank, world, local_rank = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: {world} rank(s) for --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decoder has no CPU path")
    # one rank per GPU; ENF_BENCH_SHARE_GPU=1 lets several ranks share one GPU in rehearsals (gloo)
    device = torch.device("cuda", (local_rank % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(device)

    c = CONFIGS[args.config]
    m = build_config(c, device, args.precision)
    coords = coords_of(c, c["grid"], device)
    dcoords = coords_of(c, c.get("decode_grid") or c["grid"], device)
    img = synth_targets(c, coords, c["B"], 100 + rank, device)

    if args.roofline_only:
        if world != 1:
            raise SystemExit("bench.py --roofline-only runs on one GPU")
        legs = pair_kernel_rooflines(c, m, device, args.kernel_iters)
        print(json.dumps({"roofline_kernels": {k: v for k, v in legs}, "config": {"workload": f"{c['name']}_b{c['B']}_per_gpu"}}), flush=True)
        return

    for _ in range(args.warmup):
        step(c, m, coords, dcoords, img)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, recon = step(c, m, coords, dcoords, img)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(device)
    dt = _max_over_ranks(time.perf_counter() - t0, device, world)

    pts_per_step = world * c["B"] * points_per_signal(c)
    ms_per_step = dt / args.steps * 1e3
    nd = c.get("decode_grid") or c["grid"]
    metric = "query-points/sec (ENF fit+decode) at 64 latents x 64^2 grid" if args.config == HEADLINE else \
        f"query-points/sec (ENF fit+decode), BASELINE.json config {args.config}: {c['Z']} latents x {nd[0]}x{nd[1]} grid"
    result = {
        "metric": metric,
        "value": round(pts_per_step * args.steps / dt, 1), "unit": "query-points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision if args.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"{c['name']}_b{c['B']}_per_gpu", "baseline_config": args.config, "grid": list(c["grid"]),
                   "decode_grid": list(nd), "decoded_states": c.get("rollout", 0) + 1, "latents": c["Z"],
                   "signals_per_gpu": c["B"], "inner_steps": c["S"], "sampled_points": min(c["N_s"], c["grid"][0] * c["grid"][1]),
                   "num_hidden": c["D"], "num_heads": c["H"], "latent_dim": c["C"], "num_out": c["O"], "invariant": c["inv"],
                   "hparams": c["src"], "parallelism": f"dp{world} (signals sharded, no data-path collective in fit/decode)"},
        "final_fit_loss": round(float(loss), 6),
        "timing": {"value_from": "wall clock over exactly --steps steps between barrier + synchronize, max over ranks (the contract)"},
    }
    if not args.no_meta and not c.get("rollout"):
        meta = meta_leg(c, m, coords, img, device, world)          # every rank: the outer step has the collective
        result["meta_step"] = meta
    if rank == 0:
        if args.events_steps > 0:
            result["timing"]["events"] = events_leg(c, m, coords, dcoords, img, device, args.events_steps)
        if args.graph_leg and not c.get("rollout"):
            result["timing"]["graph"] = graph_leg(c, m, coords, dcoords, img, device, max(args.events_steps, 20))
        if not args.no_accuracy:
            _, lat_fit = fit(m, coords, img)
            result["accuracy"] = accuracy_leg(c, m, dcoords, lat_fit, device)
        result["split"] = split_leg(c, m, coords, dcoords, img, device)
        result["roofline_step"] = step_roofline(c, ms_per_step, args.precision)
        if not args.no_roofline:
            legs = pair_kernel_rooflines(c, m, device, args.kernel_iters)
            result["roofline_kernels"] = {k: v for k, v in legs}
            # the dominant kernel = the one the step spends most time in (launch time x launches per step)
            dom = max(legs, key=lambda kv: kv[1]["launch_ms"] * kv[1]["launches_per_step"])
            result["roofline"] = dict(dom[1], leg=dom[0],
                                      note="dominant pair kernel of the step by time; achieved = as-written FLOPs / launch time "
                                           "(backward = 2 x forward, SURVEY.md 8d); executed_frac = MFMA FLOPs issued / time / peak; "
                                           "traffic is not measured in-run (traffic_profile = the committed PMC run, if any)")
        if world == 1 and not args.no_ode and args.config == 2:
            try:
                result["ode_eval"] = ode_leg(c, device)
            except Exception as e:                       # reported in the line, never hidden: the leg is beside the north-star metric
                result["ode_eval"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_leg(c)
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()          # ranks leave together (rank 0 has just run its extra legs)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
