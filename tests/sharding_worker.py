"""One rank of the CPU rehearsal of `bench.py --gpus N`: started by
drbamd.sharding.spawn_ranks, it reads the launcher's environment, takes its
share of the streams, converts them (with the CPU restatement: the GPU library
cannot run here), and joins the reduction bench.py reports.  Rank 0 prints one
JSON line."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "audio-resampler_amd")]

import torch.distributed as dist  # noqa: E402

import support  # noqa: E402
from drbamd import sharding  # noqa: E402


def main():
    scaling, per_gpu, total, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, local_rank, world = sharding.rank_of_environment()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    streams, first, job = sharding.streams_of_rank(scaling, per_gpu, total, rank, world)
    status, plan = support.oracle_plan(44100, 48000, 2, 4, 64, support.PUSH, support.GOOD)
    assert status == 1
    frames = 0
    for s in range(first, first + streams):
        y = support.oracle_push(plan, support.white_noise(2, 1500 + 10 * s, 900 + s))
        np.save(os.path.join(out_dir, "stream%03d.npy" % s), y)
        frames += y.shape[1]
    slowest, all_frames = sharding.reduce_step_stats(1.0 + rank, frames)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "local_rank": local_rank, "slowest": slowest, "frames": all_frames,
                          "streams": job, "first": first, "mine": streams}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
