"""Several batches and several HIP streams in one process: what the one-process-
per-GPU launch of bench.py relies on, as far as one GPU can show it."""
import os
import subprocess
import sys

import numpy as np
import pytest

import support
from support import PUSH

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
drbamd = pytest.importorskip("drbamd")

import gpu_support  # noqa: E402


def test_two_batches_side_by_side_on_their_own_streams():
    """Two batches (as two ranks would own them; here both on device 0), each fed
    on a HIP stream of its own with steps interleaved in time: each is
    bit-identical to its own converter."""
    ch, frames = 2, 30000
    status, plan = support.oracle_plan(44100, 48000, ch, 4, 64, PUSH, support.GOOD)
    config = drbamd.make_config(44100, 48000, ch, 4, 64, PUSH, support.GOOD, streams=3, device=0)
    batches = [drbamd.Batch(config), drbamd.Batch(config)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    x = [np.stack([support.white_noise(ch, frames, 70 + 10 * b + s) for s in range(3)]) for b in range(2)]
    source = [torch.from_numpy(v).cuda() for v in x]
    capacity = batches[0].query([frames] * 3)[0] + 64
    target = [torch.zeros((3, ch, capacity), device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    cuts = [0, 7001, 7002, 20000, frames]
    made = [0, 0]
    for i in range(len(cuts) - 1):
        for b in range(2):
            counts = (drbamd.ctypes.c_int * 3)()
            take = cuts[i + 1] - cuts[i]
            batches[b].process_dense(source[b].data_ptr() + 4 * cuts[i], ch * frames, frames,
                                     target[b].data_ptr() + 4 * made[b], ch * capacity, capacity,
                                     take, capacity - made[b], stream=streams[b].cuda_stream, produced=counts)
            made[b] += counts[0]
    for b in range(2):
        batches[b].synchronize(streams[b].cuda_stream)
        got = target[b].cpu().numpy()
        for s in range(3):
            want = support.oracle_push(plan, x[b][s])
            assert made[b] == want.shape[1]
            assert np.array_equal(got[s, :, :made[b]].view(np.uint32), want.view(np.uint32)), (b, s)
        batches[b].close()


def test_one_batch_fed_from_alternating_streams_keeps_its_order():
    """Steps of one batch enqueued on two different HIP streams in turn: the batch
    makes each stream wait for the step before, so history and descriptors are
    never read while the last step still writes them."""
    ch, frames = 2, 200000
    status, plan = support.oracle_plan(44100, 48000, ch, 4, 64, PUSH, support.GOOD)
    config = drbamd.make_config(44100, 48000, ch, 4, 64, PUSH, support.GOOD, streams=4)
    batch = drbamd.Batch(config)
    hip_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    x = np.stack([support.white_noise(ch, frames, 170 + s) for s in range(4)])
    source = torch.from_numpy(x).cuda()
    capacity = batch.query([frames] * 4)[0] + 64
    target = torch.zeros((4, ch, capacity), device="cuda:0")
    torch.cuda.synchronize()
    made, at, i = 0, 0, 0
    for take in gpu_support.chunk_schedule(frames, [50000, 3, 41000]):
        counts = (drbamd.ctypes.c_int * 4)()
        batch.process_dense(source.data_ptr() + 4 * at, ch * frames, frames,
                            target.data_ptr() + 4 * made, ch * capacity, capacity,
                            take, capacity - made, stream=hip_streams[i % 2].cuda_stream, produced=counts)
        at += take
        made += counts[0]
        i += 1
    for stream in hip_streams:
        batch.synchronize(stream.cuda_stream)
    got = target.cpu().numpy()
    for s in range(4):
        want = support.oracle_push(plan, x[s])
        assert made == want.shape[1]
        assert np.array_equal(got[s, :, :made].view(np.uint32), want.view(np.uint32)), s
    batch.close()


def test_streams_started_over_between_steps_on_different_streams():
    """drb_amd_batch_reset does not wait for the device: it clears the histories on the
    HIP stream of the batch's most recent step, behind that step, and the next step,
    on whatever stream, comes behind the clearing.  A long step on one stream, then
    two of the four streams started over, then a step on another stream with nothing
    waited for in between: the streams started over equal a fresh converter on the
    new input, the others one that saw both inputs, bit for bit (exact arithmetic)."""
    ch, frames = 2, 300000
    status, plan = support.oracle_plan(44100, 48000, ch, 4, 64, PUSH, support.GOOD)
    config = drbamd.make_config(44100, 48000, ch, 4, 64, PUSH, support.GOOD, streams=4)
    batch = drbamd.Batch(config)
    hip_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    first = np.stack([support.white_noise(ch, frames, 270 + s) for s in range(4)])
    second = np.stack([support.white_noise(ch, frames, 280 + s) for s in range(4)])
    sources = [torch.from_numpy(first).cuda(), torch.from_numpy(second).cuda()]
    capacity = batch.query([frames] * 4)[0] + 64
    targets = [torch.zeros((4, ch, capacity), device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    made = []
    for i in range(2):
        counts = (drbamd.ctypes.c_int * 4)()
        batch.process_dense(sources[i].data_ptr(), ch * frames, frames, targets[i].data_ptr(), ch * capacity, capacity,
                            frames, capacity, stream=hip_streams[i].cuda_stream, produced=counts)
        made.append(list(counts))
        if i == 0:
            batch.reset(1, 2)      # streams 1 and 2 start over, the step above still running
    for stream in hip_streams:
        batch.synchronize(stream.cuda_stream)
    got = targets[1].cpu().numpy()
    for s in range(4):
        if s in (1, 2):
            want = support.oracle_push(plan, second[s])
        else:
            both = support.oracle_push(plan, np.concatenate([first[s], second[s]], axis=1))
            want = both[:, made[0][s]:]
        assert made[1][s] == want.shape[1], (s, made[1][s], want.shape)
        assert np.array_equal(got[s, :, :made[1][s]].view(np.uint32), want.view(np.uint32)), s
    batch.close()


def test_host_times_count_waiting_apart_from_work():
    """drb_amd_host_times: what the host spent inside the process calls, with the time
    it waited for one of a batch's step slots (the device's time) in a bucket of
    its own; starting over clears it."""
    drbamd.host_times(start_over=True)
    config = drbamd.make_config(44100, 48000, 2, 256, 16, PUSH, support.GOOD, arithmetic=drbamd.FAST, streams=64)
    batch = drbamd.Batch(config)
    frames = 44100 * 2 // 256 * 256
    source = torch.rand((64, 2, frames), device="cuda:0") * 2 - 1
    capacity = batch.query([frames] * 64)[0] + 4096
    target = torch.zeros((64, 2, capacity), device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(12):        # more steps than the ring has slots: some of them wait
        batch.process_dense(source.data_ptr(), 2 * frames, frames, target.data_ptr(), 2 * capacity, capacity,
                            frames, capacity, stream=stream)
        batch.reset()
    batch.synchronize(stream)
    steps, spent = drbamd.host_times()
    assert steps == 12 and len(spent) == 5
    assert all(value >= 0.0 for value in spent) and sum(spent[:4]) > 0.0
    assert sum(spent[:4]) < 12 * 5000.0          # (the host's own work: far below 5 ms a step)
    assert drbamd.host_times(start_over=True)[0] == 12
    assert drbamd.host_times() == (0, [0.0] * 5)
    batch.close()


def test_a_device_that_is_not_there_is_refused():
    count = torch.cuda.device_count()
    config = drbamd.make_config(44100, 48000, 2, 4, 64, PUSH, support.GOOD, streams=1, device=count)
    with pytest.raises(drbamd.DrbError):
        drbamd.Batch(config)
    if count == 1:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        done = subprocess.run([sys.executable, os.path.join(support.ROOT, "bench.py"), "--gpus", "2"],
                              env=env, capture_output=True, text=True, timeout=120)
        assert done.returncode == 2 and "2 GPUs asked for, 1 visible" in done.stderr
