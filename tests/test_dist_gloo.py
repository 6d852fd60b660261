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
