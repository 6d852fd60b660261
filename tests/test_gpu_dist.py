"""Two ranks on one GPU (gloo, so both may share the card): the outer steps of the trainer exchange exactly one flat
all-reduce and leave every rank with identical parameters -- nef_train_step and ode_train_step, each rank
on its own shard of the meta-batch; the result equals a single process that averages the two shards' gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from enf_pde_amd.fitting import init_distributed, shard_range
    from tests.test_gpu_ode_trainer import _setup
    from tests.test_gpu_ode import _flat
    init_distributed(backend="gloo")
    cuda = torch.device("cuda:0")
    cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t = _setup(cuda)
    batch = t(traj)                                              # 2 trajectories: one per rank
    lo, hi = shard_range(batch.shape[0], rank, world)
    mk = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(1))[:32] for _ in range(3)], 1).to(cuda)
    pm = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(2))[:32] for _ in range(3)]).to(cuda)
    l1, s1 = tr.nef_train_step(state, batch[lo:hi, 0], masks=mk)
    l2, s2 = tr.ode_train_step(s1, batch[lo:hi], masks=mk, point_masks=pm)
    w = torch.cat([x.reshape(-1) for x in tr.nef.param_tensors(s2.params["nef"])]).cpu()
    o = torch.cat([v.reshape(-1) for _, v in _flat(s2.params["ode_params"])]).cpu()
    q.put((rank, float(l1), float(l2), w.numpy(), o.numpy()))       # plain arrays: no shared-memory handles to outlive us
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_outer_step(cuda):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, a1, a2, wa, oa), (_, b1, b2, wb, ob) = [(r, x, y, torch.from_numpy(w_), torch.from_numpy(o_)) for r, x, y, w_, o_ in res]
    assert a1 == b1 and a2 == b2                                 # the reported loss is the all-reduced mean
    assert torch.equal(wa, wb) and torch.equal(oa, ob)           # identical updates on every rank
    # the single-process step on the whole batch: same nef update (per-signal losses average the same way)
    from tests.test_gpu_ode_trainer import _setup
    from tests.test_gpu_ode import _flat
    cfg, prm, ocfg, oprm, coords, traj, conf, tr, state, t = _setup(cuda)
    batch = t(traj)
    mk = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(1))[:32] for _ in range(3)], 1).to(cuda)
    pm = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(2))[:32] for _ in range(3)]).to(cuda)
    l1, s1 = tr.nef_train_step(state, batch[:, 0], masks=mk)
    l2, s2 = tr.ode_train_step(s1, batch, masks=mk, point_masks=pm)
    w = torch.cat([x.reshape(-1) for x in tr.nef.param_tensors(s2.params["nef"])]).cpu()
    o = torch.cat([v.reshape(-1) for _, v in _flat(s2.params["ode_params"])]).cpu()
    assert abs(float(l1) - a1) < 1e-4 * abs(a1) and abs(float(l2) - a2) < 1e-3 * abs(a2)
    # the first Adam step is lr * sign-ish(g): entries whose gradient is within rounding / finite-difference noise of zero
    # (the step size of the Hessian-vector differences is chosen per batch) may move the other way; the rest agree
    assert ((w - wa).abs() < 2e-4).float().mean() > 0.97 and ((o - oa).abs() < 2e-4).float().mean() > 0.97


@pytest.mark.parametrize("gpus,config", [(2, 2), (4, 4)])
def test_bench_spawns_its_own_ranks(cuda, gpus, config):
    """`bench.py --gpus N` with no torchrun environment launches N ranks itself (here all on the one GPU, over gloo:
    ENF_BENCH_SHARE_GPU=1, a rehearsal switch) and the line it prints says N ranks ran, with the outer step's all-reduce
    inside the timed meta_step leg.  (4, 4): BASELINE config 4's data-parallel workload (128 latents, 128 x 128 grid, 8 signals
    per rank) at the largest rank count one GPU box admits (its process guard allows 6 GPU processes; the 8-rank exchange
    itself is covered on CPU: tests/test_dist_gloo.py)."""
    import json
    import subprocess
    import sys
    from bench import CONFIGS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ENF_BENCH_SHARE_GPU"] = "1"
    steps = 2 if gpus == 2 else 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--config", str(config), "--steps", str(steps),
                        "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == gpus and d["config"]["parallelism"].startswith(f"dp{gpus}") and d["scaling"] == "weak"
    assert d["meta_step"]["n_gpus"] == gpus and "gloo" in d["meta_step"]["collective"] and d["meta_step"]["ms_per_step"] > 0
    c = CONFIGS[config]
    one = c["B"] * ((c["S"] + 1) * c["N_s"] + c["grid"][0] * c["grid"][1])
    assert abs(d["value"] - gpus * one * steps / (d["ms_per_step"] * steps * 1e-3)) < 1e-3 * d["value"]      # the whole job's points / time


@pytest.mark.parametrize("flags", [["--config", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                                   ["--config", "3", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-meta", "--no-roofline"],
                                   ["--roofline-only"]])
def test_bench_flags(cuda, flags):
    """bench.py's other workloads and legs keep running: a BASELINE.json config other than the headline one prints the
    contract's line with its own workload name; --roofline-only prints the three per-kernel legs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + flags, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    if "--roofline-only" in flags:
        assert set(d["roofline_kernels"]) == {"decode_fwd", "fit_fwd", "fit_bwd"}
        assert all(0 < v["frac"] < 1 and v["launch_ms"] > 0 for v in d["roofline_kernels"].values())
        return
    cfg = int(flags[1])
    assert d["config"]["baseline_config"] == cfg and d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "query-points/s"
    assert f"config {cfg}" in d["metric"] and d["vs_baseline"] is None and d["higher_is_better"] is True
    if "--no-roofline" not in flags:
        assert d["roofline"]["kernel"] in ("enf_pair_fwd_kernel", "enf_pair_bwd_kernel") and d["roofline"]["traffic"] is None
        assert d["meta_step"]["ms_per_step"] > 0


def test_bench_default_line_carries_the_ode_leg(cuda):
    """The headline workload's line also reports the latent-ODE derivative evaluation (`ode_eval`): measured, not an error string."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-meta",
                        "--no-roofline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ode = d["ode_eval"]
    assert "error" not in ode, ode
    assert 0 < ode["ms_forward_graphed"] < ode["ms_forward_backward_graphed"] and ode["ms_forward_backward_eager"] > 0
    assert d["config"]["baseline_config"] == 2 and d["value"] > 0
