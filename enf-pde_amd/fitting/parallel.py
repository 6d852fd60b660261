"""Meta-batch data parallelism: one process per GPU, signals sharded evenly,
no data-path collective in fit/decode; the outer step exchanges ONE flat all-reduce (RCCL over
xGMI on GPUs, gloo in CPU tests)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("ENF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        if torch.cuda.is_available():        # every backend: a gloo run on a multi-GPU node must not pile all ranks on cuda:0
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(num_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of ``num_items`` signals for ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_(tensors, weight=None):
    """Average a list of same-dtype tensors across ranks with ONE collective on a flat buffer
    (2.1 MB of outer gradients is latency-bound on the xGMI ring: one message, not 50).

    ``weight``: this rank's share of the job (its number of signals).  Every tensor is then a per-rank MEAN over
    ``weight`` items and the result is the mean over all items of the job, sum_r w_r t_r / sum_r w_r -- the reference's
    global-batch gradient also when shard_range hands ranks shards that differ by one signal.  None = equal shards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors] + [tensors[0].new_ones(1)])
    if weight is not None:
        flat *= float(weight)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= flat[-1].clone()             # sum of the weights (the world size when unweighted)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors
