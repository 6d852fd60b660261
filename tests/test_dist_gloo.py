"""world_size-2 gloo test of the data-parallel plumbing: signals shard disjointly and
the one flat all-reduce of the outer step averages every tensor across ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from enf_pde_amd.fitting import init_distributed, shard_range, allreduce_mean_
    r, w, _ = init_distributed(backend="gloo")
    lo, hi = shard_range(13, r, w)
    grads = [torch.full((5,), float(r + 1)), torch.arange(6, dtype=torch.float32).reshape(2, 3) * (r + 1)]
    allreduce_mean_(grads)
    # unequal shards (13 signals over 2 ranks = 7 + 6): per-rank means weighted by the shard size give the job's mean
    items = torch.arange(13, dtype=torch.float32) ** 2
    wm = [items[lo:hi].mean().reshape(1)]
    allreduce_mean_(wm, weight=hi - lo)
    assert abs(wm[0].item() - items.mean().item()) < 1e-4
    # per-rank "work": each rank sums its shard; the job total must be the serial total
    part = torch.tensor([float(sum(range(lo, hi)))])
    dist.all_reduce(part)
    q.put((r, lo, hi, grads[0].tolist(), grads[1].tolist(), part.item()))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, g0a, g0b, t0), (r1, lo1, hi1, g1a, g1b, t1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 7, 7, 13)
    assert g0a == g1a == [1.5] * 5
    assert g0b == g1b == [[0.0, 1.5, 3.0], [4.5, 6.0, 7.5]]
    assert t0 == t1 == float(sum(range(13)))


def test_single_process_is_a_noop():
    from enf_pde_amd.fitting import allreduce_mean_, shard_range
    t = [torch.ones(3)]
    assert allreduce_mean_(t)[0].tolist() == [1.0, 1.0, 1.0]
    assert shard_range(10, 0, 1) == (0, 10)


def _worker8(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.set_num_threads(1)
    from enf_pde_amd.fitting import init_distributed, shard_range, allreduce_mean_
    r, w, _ = init_distributed(backend="gloo")
    lo, hi = shard_range(total, r, w)
    # per-signal "outer gradients" of three tensors of different shapes; each rank holds the MEAN over its shard, as the
    # trainer's outer step does (fitting/trainers/pde_trainer.py: nef_train_step), and one flat weighted all-reduce must give
    # the mean over all signals of the job -- also when the shards differ by one signal (62 = 6 x 8 + 2 x 7)
    g = torch.Generator().manual_seed(7)
    per_signal = [torch.randn(total, 5, generator=g), torch.randn(total, 2, 3, generator=g), torch.randn(total, 1, generator=g)]
    mine = [t[lo:hi].mean(0) for t in per_signal]
    allreduce_mean_(mine, weight=hi - lo)
    err = max(float((m - t.mean(0)).abs().max()) for m, t in zip(mine, per_signal))
    q.put((r, lo, hi, err))
    dist.barrier()
    dist.destroy_process_group()


def _run8(total):
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_eight_rank_gloo_weighted_allreduce_equal_and_unequal_shards():
    """BASELINE config 4's exchange step at its rank count (meta-batch 64 over 8 ranks), on CPU / gloo: 64 signals
    give 8 x 8 shards, 62 give uneven ones; the single weighted all-reduce reproduces the global-batch mean in both."""
    res = _run8(64)
    assert [(lo, hi) for _, lo, hi, _ in res] == [(8 * r, 8 * r + 8) for r in range(8)]
    assert max(e for *_, e in res) < 1e-5
    res = _run8(62)
    sizes = [hi - lo for _, lo, hi, _ in res]
    assert sizes == [8] * 6 + [7] * 2 and res[0][1] == 0 and res[-1][2] == 62
    assert all(res[i][2] == res[i + 1][1] for i in range(7))          # contiguous, disjoint
    assert max(e for *_, e in res) < 1e-5
