"""Stream sharding across the GPUs of one node.

Streams are independent converters, so a job over `total` streams is split into
contiguous stream-id ranges, one per rank (one process per GPU); no data moves
between ranks.  torch.distributed is used only to agree on timings."""


def shard_range(total, rank, world):
    """Contiguous [first, last) of stream ids owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def reduce_step_stats(elapsed_seconds, frames, device=None):
    """(slowest rank's time, frames summed over ranks); identity without a process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_seconds), float(frames)
    slowest = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device)
    total = torch.tensor([frames], dtype=torch.float64, device=device)
    dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return slowest.item(), total.item()
