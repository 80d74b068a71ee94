"""The four-function drop-in (drb_audio_converter_*) running on the GPU,
driven by the same harness that drives the upstream build: same sample
stream, same callback frame counts, same example program output."""
import numpy as np
import pytest

import golden_data
import support
from support import PUSH, PULL

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
drbamd = pytest.importorskip("drbamd")

META, DATA = golden_data.streams()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def harness():
    return support.ApiHarness(drbamd.HARNESS_PATH)


NAMES = ["cfg1_push_good", "cfg1_pull_good", "cfg2_push_good", "cfg3_push_good", "cfg4_pull_good",
         "cfg5_push_best", "cfg1_push_good_bs1", "same_rate_reblock", "up2_push_bs1"]


@pytest.mark.parametrize("name", NAMES)
def test_drop_in_stream_and_callbacks(harness, name):
    meta = META[name]
    args = (meta["src"], meta["dst"], meta["channels"], meta["block_size"], meta["max_block_count"], meta["quality"])
    x = DATA["%s.noise.in" % name]
    want = DATA["%s.noise.out" % name]
    if meta["direction"] == PUSH:
        got, _ = harness.push(*args, x, [x.shape[1]])
    else:
        got, _, consumed = harness.pull(*args, x, [want.shape[1]])
        assert consumed == meta["used"]
    assert got.shape == want.shape
    assert np.array_equal(bits(got), bits(want))
    for sched in meta["callbacks"].values():
        big = support.white_noise(meta["channels"], 200000, 3)
        if meta["direction"] == PUSH:
            _, counts = harness.push(*args, big, sched["chunks"])
        else:
            _, counts, _ = harness.pull(*args, big, sched["chunks"])
        assert counts.tolist() == sched["counts"]


def test_sizes_match_upstream(harness):
    for row in golden_data.planner_rows()[::97]:
        if row[7] in ("rejected", "unsupported"):
            assert harness.sizes(*row[:7]) == (0, 0)
            continue
        status, plan = support.oracle_plan(*row[:7])
        if status != 1 or not plan.consistent:
            continue
        assert harness.sizes(*row[:7]) == (row[11], row[12]), row


def format_example(frames):
    return "".join("%3d" % i + "".join(", %8.3f" % v for v in frames[:, i]) + "\n"
                   for i in range(frames.shape[1]))


def test_example_push_program_output(harness):
    slices = [13, 17, 4, 7, 5, 4, 21, 29, 300]
    x = np.stack([np.arange(500) + 100.0 * c for c in range(2)]).astype(np.float32)
    y, counts = harness.push(44100, 48000, 2, 4, 64, support.GOOD, x, slices)
    assert all(c % 4 == 0 for c in counts)
    assert format_example(y) == golden_data.example_text("example_push")


def test_example_pull_program_output(harness):
    slices = [13, 17, 4, 7, 5, 4, 21, 29, 300]
    x = np.stack([np.arange(4000) + 100.0 * c for c in range(2)]).astype(np.float32)
    y, counts, _ = harness.pull(44100, 48000, 2, 4, 256, support.GOOD, x, slices)
    assert all(c % 4 == 0 for c in counts)
    want = golden_data.example_text("example_pull").splitlines()[:sum(slices)]
    assert format_example(y).splitlines() == want


def test_callback_timestamps_match_upstream(harness):
    """Upstream hands `timestamp + latency` to the data callback, where latency
    is what its passes' frame-count functions leave behind (half-bands and the
    slicer add seconds, the sinc stage overwrites the sum with frames:
    drb-audio-converter.c:1521, 1540, 1948, 1962, 2497, 3576, 3639).  The
    figure is replicated bit for bit; the fixture comes from the upstream build
    (tests/golden/make_stamps.py)."""
    import json

    with open(support.ROOT + "/tests/golden/callback_stamps.json") as handle:
        rows = json.load(handle)
    ran = 0
    for name, row in rows.items():
        src, dst, ch, bs, mbc, direction, quality = row["args"]
        x = support.white_noise(ch, 200000, 5)
        if direction == PUSH:
            result = harness.push(src, dst, ch, bs, mbc, quality, x, row["chunks"], timestamp=row["timestamp"])
        else:
            result = harness.pull(src, dst, ch, bs, mbc, quality, x, row["chunks"], timestamp=row["timestamp"])
        if result is None:
            continue   # a conversion this library declines
        ran += 1
        assert result[1].tolist() == row["counts"], name
        assert [float(v).hex() for v in harness.stamps] == row["stamps"], name
    assert ran >= 5


def test_converters_that_are_never_released_do_not_run_into_the_mapping_limit():
    """Upstream has no destructor, so a program written against its API never releases a
    converter; this library keeps each one's device state until the process ends.
    Every pinned host allocation is a memory mapping, of which a process has about
    65 000: with nine of them per converter construction failed ("out of memory",
    on an all but empty GPU) at converter 7 272.  9 000 converters are
    constructed here in blocks of their own with none released in between; then
    all are released."""
    import ctypes

    lib = drbamd.lib

    class Callback(ctypes.Structure):
        _fields_ = [("process", ctypes.c_void_p), ("state", ctypes.c_void_p)]

    lib.drb_audio_converter_alignment_and_size.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_long)] * 2
    lib.drb_audio_converter_alignment_and_size.restype = ctypes.c_bool
    lib.drb_audio_converter_construct.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [Callback]
    lib.drb_audio_converter_construct.restype = ctypes.c_void_p
    lib.drb_amd_converter_release.argtypes = [ctypes.c_void_p]
    lib.drb_amd_converter_release.restype = None

    request = (44100, 48000, 2, 256, 16, PUSH, support.GOOD)
    alignment, size = ctypes.c_long(), ctypes.c_long()
    assert lib.drb_audio_converter_alignment_and_size(*request, ctypes.byref(alignment), ctypes.byref(size))
    count, each = 9000, (size.value + 63) // 64 * 64
    arena = ctypes.create_string_buffer(count * each + 64)
    base = (ctypes.addressof(arena) + 63) // 64 * 64
    converters = []
    try:
        for i in range(count):
            converter = lib.drb_audio_converter_construct(base + i * each, *request, Callback(None, None))
            assert converter, "converter %d: %s" % (i + 1, drbamd.lib.drb_amd_last_error().decode())
            converters.append(converter)
    finally:
        for converter in converters:
            lib.drb_amd_converter_release(converter)
    assert len(converters) == count
