"""Stream sharding across the GPUs of one node.

Streams are independent converters, so a job over `total` streams is split into
contiguous stream-id ranges, one per rank (one process per GPU); no data moves
between ranks.  torch.distributed is used only to agree on timings."""
import os
import socket
import subprocess
import sys
import time


def spawn_ranks(world, script, argv, check_devices=True):
    """Starts `world` copies of `script argv`, one per rank, with the rendezvous a
    launcher would set (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR 127.0.0.1, a free
    MASTER_PORT).  The caller must not have touched a GPU yet: the children are
    new processes, nothing is re-executed.  Their output passes through; returns
    0 if every rank ended with 0, else 1 (the other ranks are ended as soon as
    one fails)."""
    if check_devices:
        import torch

        have = torch.cuda.device_count()   # (counts devices without initialising any)
        if have < world:
            sys.stderr.write("%d GPUs asked for, %d visible\n" % (world, have))
            return 2
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    children = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
    # The ranks meet at barriers: when one dies the others would sit there until
    # the collective's timeout.  So the children are polled, and as soon as one
    # ends badly the rest are ended too (the very processes started above, by
    # their own handles).  A rank killed by a signal counts as 128 + signal.
    worst = 0
    running = list(children)
    while running:
        for child in list(running):
            code = child.poll()
            if code is None:
                continue
            running.remove(child)
            if code != 0:
                worst = max(worst, 128 - code if code < 0 else code)
                for other in running:
                    other.terminate()
        if running:
            time.sleep(0.05)
    for child in children:
        if child.returncode is None:
            try:
                child.wait(timeout=10)
            except subprocess.TimeoutExpired:
                child.kill()
                child.wait()
    return 1 if worst else 0


def rank_of_environment():
    """(rank, local rank, world size) as a launcher set them; (0, 0, 1) outside one."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def streams_of_rank(scaling, per_gpu, total, rank, world):
    """(streams this rank owns, first stream id, streams of the whole job).  weak:
    every rank owns `per_gpu` streams; strong: `total` streams (0 = 8 x per_gpu) are
    split into contiguous ranges."""
    if scaling == "strong":
        total = total or 8 * per_gpu
        first, last = shard_range(total, rank, world)
        return last - first, first, total
    return per_gpu, rank * per_gpu, per_gpu * world


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
