"""The entry point that allocates nothing, drb_amd_batch_process_dense_static,
against drb_amd_batch_process_dense on the same inputs: two batches of one
configuration, fed the same chunks, one through each entry point; two calls in
a row, the second with a smaller input.  Both run the same kernels on the same
data, so the outputs are compared bit for bit (which meets the parity bar of
either arithmetic mode: the regular entry point's outputs are checked against
the upstream converter by the other tests).  Around every static call the
free device memory (hipMemGetInfo) and the library's own count of its
allocation calls (host, device, pinned, streams, events) are read: both must
be what they were."""
import ctypes

import numpy as np
import pytest

import support
from support import PUSH

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
drbamd = pytest.importorskip("drbamd")

CASES = [
    ("exact", drbamd.EXACT, 44100, 48000, 2, support.GOOD, 5),
    ("fast, split matrix-core kernel", drbamd.FAST, 44100, 48000, 2, support.GOOD, 6),
    ("fast, one decimator", drbamd.FAST, 96000, 48000, 8, support.GOOD, 3),
    ("fast, interpolators folded into the split kernel", drbamd.FAST, 16000, 48000, 1, support.GOOD, 5),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_static_entry_point_matches_the_regular_one(case):
    _, arithmetic, src, dst, ch, quality, streams = case
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, arithmetic=arithmetic, streams=streams)
    regular, static = drbamd.Batch(config), drbamd.Batch(config)
    calls = [3 * src // 256 * 256, 4096]   # the second call is the smaller one
    total = sum(calls)
    x = np.stack([support.white_noise(ch, total, 500 + s) for s in range(streams)])
    source = torch.from_numpy(x).cuda()
    capacity = regular.query([total] * streams)[0] + 4096
    outputs = [torch.zeros((streams, ch, capacity), device="cuda:0") for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    made = [0, 0]
    at = 0
    for frames in calls:
        for which, (batch, entry) in enumerate(((regular, regular.process_dense), (static, static.process_dense_static))):
            counts = (ctypes.c_int * streams)()
            if which == 1:
                # the property itself: device memory as it was, no allocation call of the library's
                torch.cuda.synchronize()
                free_before = torch.cuda.mem_get_info()[0]
                calls_before = drbamd.lib.drb_amd_allocation_calls()
            entry(source.data_ptr() + 4 * at, ch * total, total,
                  outputs[which].data_ptr() + 4 * made[which], ch * capacity, capacity,
                  frames, capacity - made[which], stream=stream, produced=counts)
            if which == 1:
                calls_after = drbamd.lib.drb_amd_allocation_calls()
                torch.cuda.synchronize()
                assert torch.cuda.mem_get_info()[0] == free_before, "device memory allocated by the static entry point"
                assert calls_after == calls_before, "the static entry point allocated or freed something"
            assert len(set(counts)) == 1
            made[which] += counts[0]
        at += frames
    regular.synchronize(stream)
    static.synchronize(stream)
    assert made[0] == made[1] > 0
    want, got = outputs[0].cpu().numpy(), outputs[1].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if arithmetic == drbamd.EXACT:
        status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, quality)
        for s in range(streams):
            upstream = support.oracle_push(plan, x[s])
            assert np.array_equal(got[s, :, :made[1]].view(np.uint32), upstream.view(np.uint32))
    regular.close()
    static.close()


CHAINS = [
    ("three launches going down, planar float32", 48000, 16000, 2, support.GOOD, False, 3),
    ("two decimators between two launches that convert, 8 channels interleaved", 192000, 48000, 8, support.BEST, True, 4),
    ("three launches going up between two that convert, 3 channels interleaved", 11025, 48000, 3, support.FINE, True, 5),
]


@pytest.mark.parametrize("case", CHAINS, ids=[c[0] for c in CHAINS])
def test_static_entry_point_on_chains_of_several_launches_from_the_first_call(case):
    """A chain of several launches keeps planar float32 buffers between them, which grow with
    the steps.  A batch has them from its creation for steps of up to 16 384 frames, so
    that the entry point that may not allocate serves such a chain from its first
    call on: two calls of 8192 and 4096 frames on a fresh batch, bit for bit what
    the regular entry point makes, with nothing allocated."""
    _, src, dst, ch, quality, interleaved, launches = case
    streams = 3
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, arithmetic=drbamd.FAST, streams=streams,
                                layout=drbamd.INTERLEAVED if interleaved else drbamd.PLANAR)
    assert drbamd.plan_info(config).fast_launches == launches
    regular, static = drbamd.Batch(config), drbamd.Batch(config)
    calls = [8192, 4096]
    total = sum(calls)
    x = np.stack([support.white_noise(ch, total, 520 + s) for s in range(streams)])       # [stream][channel][frame]
    host = np.ascontiguousarray(x.transpose(0, 2, 1)) if interleaved else x
    source = torch.from_numpy(host).cuda()
    capacity = regular.query([total] * streams)[0] + 4096
    shape = (streams, capacity, ch) if interleaved else (streams, ch, capacity)
    outputs = [torch.zeros(shape, device="cuda:0") for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    frame_bytes = 4 * ch if interleaved else 4
    source_strides = (total * ch, 0 if interleaved else total)
    target_strides = (capacity * ch, 0 if interleaved else capacity)
    made, at = [0, 0], 0
    for frames in calls:
        for which, (batch, entry) in enumerate(((regular, regular.process_dense), (static, static.process_dense_static))):
            counts = (ctypes.c_int * streams)()
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            calls_before = drbamd.lib.drb_amd_allocation_calls()
            entry(source.data_ptr() + frame_bytes * at, *source_strides,
                  outputs[which].data_ptr() + frame_bytes * made[which], *target_strides,
                  frames, capacity - made[which], stream=stream, produced=counts)
            calls_after = drbamd.lib.drb_amd_allocation_calls()
            torch.cuda.synchronize()
            if which == 1:
                assert torch.cuda.mem_get_info()[0] == free_before, "device memory allocated by the static entry point"
                assert calls_after == calls_before, "the static entry point allocated or freed something"
            assert len(set(counts)) == 1
            made[which] += counts[0]
        at += frames
    regular.synchronize(stream)
    static.synchronize(stream)
    assert made[0] == made[1] > 0
    want, got = outputs[0].cpu().numpy(), outputs[1].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, quality)
    for s in range(streams):
        upstream = support.oracle_push(plan, x[s])
        mine = got[s, :made[1], :].T if interleaved else got[s, :, :made[1]]
        assert mine.shape == upstream.shape
        assert np.abs(mine.astype(np.float64) - upstream).max() <= 1e-6 + 1e-5 * np.abs(upstream).max()
    regular.close()
    static.close()


def test_static_entry_point_rejects_what_does_not_fit():
    """More step descriptors than the workspace holds: a clear error, nothing enqueued."""
    streams = 30000
    config = drbamd.make_config(44100, 48000, 1, 256, 16, PUSH, support.POOR, streams=streams)
    batch = drbamd.Batch(config)
    source = torch.zeros((streams, 1, 256), device="cuda:0")
    target = torch.zeros((streams, 1, 1024), device="cuda:0")
    with pytest.raises(drbamd.DrbError) as error:
        batch.process_dense_static(source.data_ptr(), 256, 256, target.data_ptr(), 1024, 1024, 256, 1024)
    assert "static workspace" in str(error.value)
    batch.close()
