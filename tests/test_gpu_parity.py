"""Parity of the HIP path with the upstream converter, through the C ABI.

Exact arithmetic must reproduce the golden vectors (produced by the upstream
scalar build) and the CPU restatement BIT FOR BIT, for any way of cutting the
stream into steps and tiles."""
import ctypes

import numpy as np
import pytest

import golden_data
import support
from support import PUSH, PULL

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
drbamd = pytest.importorskip("drbamd")

import gpu_support  # noqa: E402

META, DATA = golden_data.streams()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def config_for(meta, **kw):
    return drbamd.make_config(meta["src"], meta["dst"], meta["channels"], meta["block_size"],
                              meta["max_block_count"], meta["direction"], meta["quality"], **kw)


def unsupported(meta):
    status, plan = support.oracle_plan(meta["src"], meta["dst"], meta["channels"], meta["block_size"],
                                       meta["max_block_count"], meta["direction"], meta["quality"])
    return status != 1 or not plan.consistent


RUNNABLE = [name for name in sorted(META) if not unsupported(META[name])]


@pytest.mark.parametrize("name", RUNNABLE)
def test_golden_streams_bit_exact(name):
    """Three streams of one batch get the same data cut three different ways."""
    meta = META[name]
    for kind in meta["signals"]:
        x = DATA["%s.%s.in" % (name, kind)]
        want = DATA["%s.%s.out" % (name, kind)]
        total = x.shape[1] if meta["direction"] == PUSH else want.shape[1]
        schedules = [gpu_support.chunk_schedule(total, p)
                     for p in ([total], [13, 17, 4, 7, 5, 4, 21, 29, 300, 1], [997, 1, 2048])]
        steps = max(len(s) for s in schedules)
        runner = gpu_support.StreamRunner(config_for(meta, streams=3), [x, x, x])
        for i in range(steps):
            runner.step([s[i] if i < len(s) else 0 for s in schedules])
        for s in range(3):
            got = runner.result(s)
            assert got.shape[1] >= want.shape[1] if meta["direction"] == PUSH else got.shape == want.shape
            got = got[:, :want.shape[1]]
            assert got.shape == want.shape, (name, kind, s, got.shape, want.shape)
            assert np.array_equal(bits(got), bits(want)), (name, kind, s)
        runner.close()


@pytest.mark.parametrize("name", RUNNABLE)
def test_step_frame_counts_match_upstream_callbacks(name):
    """Steps no larger than the throttling see upstream's callback counts."""
    meta = META[name]
    info = drbamd.plan_info(config_for(meta))
    sched = meta["callbacks"]["example"]
    chunks, counts = sched["chunks"], sched["counts"]
    # cut every chunk at the throttling, as the upstream driver does
    steps = []
    for c in chunks:
        steps += gpu_support.chunk_schedule(c, [info.throttling]) if c else []
    x = support.white_noise(meta["channels"], 200000, 3)
    runner = gpu_support.StreamRunner(config_for(meta), [x])
    got = [runner.step([s])[0] for s in steps]
    runner.close()
    assert got == counts, name


@pytest.mark.parametrize("sample,layout", [(drbamd.I16, drbamd.PLANAR), (drbamd.I16, drbamd.INTERLEAVED),
                                           (drbamd.I32, drbamd.INTERLEAVED), (drbamd.F32, drbamd.INTERLEAVED)])
@pytest.mark.parametrize("name", ["cfg1_push_good", "cfg2_push_good", "cfg3_push_good", "cfg4_pull_good",
                                  "cfg5_push_best", "cfg1_push_poor", "r11025_to_60k_pull_fine"])
def test_integer_and_interleaved_buffers(name, sample, layout):
    """int16 / int32 / interleaved I-O: load conversion -> upstream float path
    -> store conversion, checked against the CPU restatement (exact ints)."""
    meta = META[name]
    ch = meta["channels"]
    status, plan = support.oracle_plan(meta["src"], meta["dst"], ch, meta["block_size"],
                                       meta["max_block_count"], meta["direction"], meta["quality"])
    assert status == 1
    frames = 3000
    noise = support.white_noise(ch, frames, 11)
    if sample == drbamd.I16:
        x = np.round(noise * 32767).astype(np.int16)
        xf = support.oracle_convert_samples(x, "f32")
    elif sample == drbamd.I32:
        x = np.round(noise.astype(np.float64) * 2147483647).astype(np.int32)
        xf = support.oracle_convert_samples(x, "f32")
    else:
        x, xf = noise, noise
    if meta["direction"] == PUSH:
        yf = support.oracle_push(plan, xf)
        total = frames
    else:
        total = int(frames * meta["dst"] / meta["src"]) - 2 * 2048
        yf, _ = support.oracle_pull(plan, xf, total)
    want = {drbamd.I16: lambda v: support.oracle_convert_samples(v, "i16"),
            drbamd.I32: lambda v: support.oracle_convert_samples(v, "i32"),
            drbamd.F32: lambda v: v}[sample](np.ascontiguousarray(yf))
    runner = gpu_support.StreamRunner(config_for(meta, sample_type=sample, layout=layout), [x])
    for step in gpu_support.chunk_schedule(total, [701, 64, 1500]):
        runner.step([step])
    got = runner.result(0)[:, :want.shape[1]]
    runner.close()
    assert got.shape == want.shape
    assert np.array_equal(got, want) if sample != drbamd.F32 else np.array_equal(bits(got), bits(want))


def test_ragged_batch_of_streams():
    """Many streams, each with its own length and its own step pattern."""
    rng = np.random.default_rng(5)
    n, ch = 37, 1
    status, plan = support.oracle_plan(16000, 48000, ch, 256, 16, PULL, support.GOOD)
    assert status == 1
    config = drbamd.make_config(16000, 48000, ch, 256, 16, PULL, support.GOOD, streams=n)
    inputs = [support.white_noise(ch, 6000, 100 + s) for s in range(n)]
    totals = [int(v) for v in rng.integers(0, 9000, n)]
    runner = gpu_support.StreamRunner(config, inputs)
    left = list(totals)
    while any(left):
        steps = [min(l, int(rng.integers(0, 4000))) for l in left]
        runner.step(steps)
        left = [l - s for l, s in zip(left, steps)]
    for s in range(n):
        want, _ = support.oracle_pull(plan, inputs[s], totals[s])
        got = runner.result(s)
        assert got.shape == want.shape, s
        assert np.array_equal(bits(got), bits(want)), s
    runner.close()


def test_reset_returns_streams_to_initial_state():
    meta = META["cfg1_push_good"]
    x = DATA["cfg1_push_good.noise.in"]
    want = DATA["cfg1_push_good.noise.out"]
    runner = gpu_support.StreamRunner(config_for(meta, streams=2), [x, x])
    runner.step([1234, 999])
    runner.batch.reset()
    runner.read = [0, 0]
    runner.outputs = [[], []]
    runner.step([x.shape[1]] * 2)
    for s in range(2):
        assert np.array_equal(bits(runner.result(s)), bits(want))
    runner.close()


def test_rejected_and_unsupported_requests():
    with pytest.raises(drbamd.DrbError) as error:
        drbamd.Batch(drbamd.make_config(44100, 48000, 2, 3, 64, PUSH, support.GOOD))
    assert error.value.status == "invalid_conversion"
    with pytest.raises(drbamd.DrbError) as error:
        drbamd.Batch(drbamd.make_config(96000, 48000, 2, 1, 64, PUSH, support.GOOD))
    assert error.value.status == "unsupported"
    batch = drbamd.Batch(drbamd.make_config(44100, 48000, 2, 4, 64, PUSH, support.GOOD))
    buf = torch.zeros(4096, device="cuda:0")
    with pytest.raises(drbamd.DrbError) as error:  # target buffer too small
        batch.process([buf.data_ptr()], 1000, [1000], [buf.data_ptr()], 8, [8])
    assert error.value.status == "capacity"
    # the rejected call changed nothing
    assert batch.query([1000]) == [support.oracle_lib().drb_oracle_push_count(
        support.oracle_plan(44100, 48000, 2, 4, 64, PUSH, support.GOOD)[1], 1000)]
    batch.close()


def test_host_fed_batch_matches_device_resident_batch():
    """Sources and targets in (pinned) host memory, handed over in slices whose
    copies overlap the conversion: same bits as the device-resident entry,
    over two steps so that state carried between steps is covered too."""
    streams, ch = 37, 2
    frames = [6000, 3100]
    config = drbamd.make_config(44100, 48000, ch, 64, 16, PUSH, support.GOOD, streams=streams)
    x = np.stack([support.white_noise(ch, sum(frames), 100 + s) for s in range(streams)])
    want = []
    for s in range(0, streams, 12):
        status, plan = support.oracle_plan(44100, 48000, ch, 64, 16, PUSH, support.GOOD)
        want.append((s, support.oracle_push(plan, x[s])))
    batch = drbamd.Batch(config)
    got = [[] for _ in range(streams)]
    offset = 0
    for count in frames:
        capacity = batch.query([count] * streams)[0] + 64
        source = torch.from_numpy(np.ascontiguousarray(x[:, :, offset:offset + count])).pin_memory()
        target = torch.zeros((streams, ch, capacity), dtype=torch.float32).pin_memory()
        produced = (ctypes.c_int * streams)()
        batch.process_host_dense(source.data_ptr(), ch * count, count, target.data_ptr(), ch * capacity, capacity,
                                 count, capacity, produced=produced)
        for s in range(streams):
            got[s].append(target[s, :, :produced[s]].numpy().copy())
        offset += count
    batch.close()
    for s, expected in want:
        result = np.concatenate(got[s], axis=1)
        assert result.shape[1] <= expected.shape[1]
        assert np.array_equal(result.view(np.uint32), expected[:, :result.shape[1]].view(np.uint32))
