"""The matrix-core ("fast") arithmetic mode against the upstream converter.

Bound from the task statement: float32 within 1e-5 relative / 1e-6 absolute of
the upstream C implementation (|got - want| <= 1e-6 + 1e-5 |want|) on inputs in
[-1, 1]; integer outputs within +-1 LSB."""
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

ABS_TOL, REL_TOL = 1e-6, 1e-5


def config_for(meta, **kw):
    return drbamd.make_config(meta["src"], meta["dst"], meta["channels"], meta["block_size"],
                              meta["max_block_count"], meta["direction"], meta["quality"],
                              arithmetic=drbamd.FAST, **kw)


def assert_close(got, want, what):
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    excess = np.abs(got - want) - (ABS_TOL + REL_TOL * np.abs(want))
    worst = int(np.argmax(excess))
    assert excess.max() <= 0, "%s: |err| %.3g at %d (want %.6g)" % (
        what, np.abs(got - want).reshape(-1)[worst], worst, want.reshape(-1)[worst])
    return float(np.abs(got - want).max())


FAST_NAMES = ["cfg1_push_good", "cfg1_pull_good", "cfg1_push_fine", "cfg1_push_best",
              "cfg1_pull_best", "cfg1_push_good_bs1", "cfg2_push_good", "cfg1_push_poor", "cfg3_push_good",
              "down2_pull_bs1", "up2_push_bs1", "up2_push_slicer"]


@pytest.mark.parametrize("name", FAST_NAMES)
def test_fast_path_is_selected(name):
    assert drbamd.plan_info(config_for(META[name])).fast_path == 1


@pytest.mark.parametrize("name", sorted(META))
def test_fast_mode_within_tolerance_of_golden(name):
    meta = META[name]
    try:
        drbamd.plan_info(config_for(meta))
    except drbamd.DrbError:
        pytest.skip("not run by this library")
    for kind in meta["signals"]:
        x = DATA["%s.%s.in" % (name, kind)]
        want = DATA["%s.%s.out" % (name, kind)]
        if kind == "ramp":
            # values up to 600, outside the [-1, 1] range the absolute floor is stated for:
            # the relative bound alone, on the scale of the signal
            scale = float(np.abs(want).max())
            runner = gpu_support.StreamRunner(config_for(meta, streams=1), [x])
            total = x.shape[1] if meta["direction"] == PUSH else want.shape[1]
            runner.step([total])
            got = runner.result(0)[:, :want.shape[1]].astype(np.float64)
            runner.close()
            assert np.abs(got - want).max() <= REL_TOL * scale, "%s.ramp: %.3g on a scale of %.3g" % (
                name, np.abs(got - want).max(), scale)
            continue
        total = x.shape[1] if meta["direction"] == PUSH else want.shape[1]
        schedules = [gpu_support.chunk_schedule(total, p) for p in ([total], [997, 1, 2048, 333])]
        steps = max(len(s) for s in schedules)
        runner = gpu_support.StreamRunner(config_for(meta, streams=2), [x, x])
        for i in range(steps):
            runner.step([s[i] if i < len(s) else 0 for s in schedules])
        for s in range(2):
            got = runner.result(s)[:, :want.shape[1]]
            assert got.shape == want.shape
            assert_close(got, want, "%s.%s stream %d" % (name, kind, s))
        runner.close()


@pytest.mark.parametrize("stride,split", [("", "0"), ("1", "0"), ("", "1"), ("1", "1")])
def test_fast_mode_long_stream_many_tiles(stride, split, monkeypatch):
    """Several tiles per step and several steps: 4 s of noise and of sweep; with
    the launcher's choice of how workgroups share a stream's tiles and with one
    workgroup walking over all of them (which hands each tile's tail on to the
    next instead of warming every tile up); with the three stages kept apart and
    as one filter on the half-precision matrix instruction."""
    if stride:
        monkeypatch.setenv("DRB_AMD_FAST_TILE_STRIDE", stride)
    monkeypatch.setenv("DRB_AMD_FAST_SPLIT", split)
    for signal in ("noise", "sweep"):
        ch, frames = 2, 4 * 44100
        x = support.white_noise(ch, frames, 21) if signal == "noise" else support.log_sweep(ch, frames, 44100, 0.45 * 44100)
        status, plan = support.oracle_plan(44100, 48000, ch, 256, 16, PUSH, support.GOOD)
        want = support.oracle_push(plan, x)
        config = drbamd.make_config(44100, 48000, ch, 256, 16, PUSH, support.GOOD, arithmetic=drbamd.FAST, streams=2)
        runner = gpu_support.StreamRunner(config, [x, x])
        schedules = [gpu_support.chunk_schedule(frames, p) for p in ([frames], [44100, 4096, 30000])]
        for i in range(max(len(s) for s in schedules)):
            runner.step([s[i] if i < len(s) else 0 for s in schedules])
        for s in range(2):
            got = runner.result(s)
            assert got.shape == want.shape
            worst = assert_close(got, want, "%s stream %d" % (signal, s))
            # staged: 4.8e-7 with coefficients rebuilt for every tile, 7.2e-7 following the
            # drift over 16; split: 9.5e-7 (one sum of ~100 products per frame)
            assert worst < (1.1e-6 if split == "1" else 8e-7)
        runner.close()


@pytest.mark.parametrize("sample,layout", [(drbamd.I16, drbamd.INTERLEAVED), (drbamd.I32, drbamd.PLANAR)])
def test_fast_mode_integer_outputs_within_one_lsb(sample, layout):
    meta = META["cfg2_push_good"]
    ch, frames = 2, 20000
    noise = support.white_noise(ch, frames, 31)
    status, plan = support.oracle_plan(48000, 44100, ch, 256, 16, PUSH, support.GOOD)
    if sample == drbamd.I16:
        x = np.round(noise * 32767).astype(np.int16)
        to = "i16"
    else:
        x = np.round(noise.astype(np.float64) * (2 ** 24 - 1)).astype(np.int32) * 128
        to = "i32"
    xf = support.oracle_convert_samples(x, "f32")
    want_f = support.oracle_push(plan, xf)
    want = support.oracle_convert_samples(np.ascontiguousarray(want_f), to)
    runner = gpu_support.StreamRunner(config_for(meta, sample_type=sample, layout=layout), [x])
    for step in gpu_support.chunk_schedule(frames, [7001, 640, 15000]):
        runner.step([step])
    got = runner.result(0)
    runner.close()
    assert got.shape == want.shape
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    if sample == drbamd.I16:
        assert diff.max() <= 1
    else:
        # One step of 2^-31 is finer than float32 resolves: only upstream's own float32
        # bit for bit is within it, so fast arithmetic on int32 buffers runs the exact
        # kernels (the header says so) and the integers are identical.
        assert drbamd.plan_info(config_for(meta, sample_type=sample, layout=layout)).fast_path == 0
        assert diff.max() == 0


def test_fast_mode_whole_ten_second_stream_per_step():
    """The benchmark's step: a whole 10 s stream in one call (188 tiles per
    channel, 24 stretches of coefficient sets), with the kernels the library
    picks by default; the second stream takes the same samples in two calls of
    odd lengths, so that its phases, and with them its coefficient sets, differ
    from the first stream's in the second call."""
    ch, frames = 2, 10 * 44100
    x = support.white_noise(ch, frames, 77)
    status, plan = support.oracle_plan(44100, 48000, ch, 256, 16, PUSH, support.GOOD)
    assert status == 1
    want = support.oracle_push(plan, x)
    config = drbamd.make_config(44100, 48000, ch, 256, 16, PUSH, support.GOOD, arithmetic=drbamd.FAST, streams=2)
    assert drbamd.plan_info(config).fast_collapsed == 2   # the split form of the collapsed filter
    runner = gpu_support.StreamRunner(config, [x, x])
    schedules = [gpu_support.chunk_schedule(frames, p) for p in ([frames], [12345, frames - 12345])]
    for i in range(max(len(s) for s in schedules)):
        runner.step([s[i] if i < len(s) else 0 for s in schedules])
    for s in range(2):
        got = runner.result(s)
        assert got.shape == want.shape
        assert_close(got, want, "stream %d" % s)
    runner.close()


FOLDED = [
    # (source rate, target rate, channels, quality, 1 where the interpolators fold into one launch, frames)
    (16000, 48000, 1, support.GOOD, 1, 3 * 16000),
    (8000, 48000, 2, support.GOOD, 1, 2 * 8000),
    (16000, 96000, 2, support.GOOD, 1, 2 * 16000),
    (22050, 96000, 2, support.FINE, 0, 2 * 22050),    # 147 : 160 behind three interpolators: no segment of whole periods fits
]


@pytest.mark.parametrize("src,dst,ch,quality,folds,frames", FOLDED)
@pytest.mark.parametrize("fold", ["", "0"])
def test_fast_mode_chains_of_interpolators(src, dst, ch, quality, folds, frames, fold, monkeypatch):
    """Interpolators in front of the fractional stage's launch folded into its
    filter (one launch, no intermediate rates in memory), and, as a
    diagnostic, kept apart: both against the oracle over several tiles and
    calls."""
    if fold:
        monkeypatch.setenv("DRB_AMD_FAST_FOLD", fold)
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, quality)
    if status != 1:
        pytest.skip("not a conversion upstream accepts")
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, arithmetic=drbamd.FAST, streams=2)
    info = drbamd.plan_info(config)
    if not info.fast_path:
        pytest.skip("no matrix-core kernel for this chain")
    if not fold and folds:
        assert info.fast_launches == 1 and info.fast_collapsed == 2
    x = support.white_noise(ch, frames, 33)
    want = support.oracle_push(plan, x)
    runner = gpu_support.StreamRunner(config, [x, x])
    schedules = [gpu_support.chunk_schedule(frames, p) for p in ([frames], [frames // 3, 4096, 777])]
    for i in range(max(len(s) for s in schedules)):
        runner.step([s[i] if i < len(s) else 0 for s in schedules])
    for s in range(2):
        got = runner.result(s)
        assert got.shape == want.shape
        assert_close(got, want, "%d -> %d stream %d" % (src, dst, s))
    runner.close()


RAGGED = [
    # (source rate, target rate, channels, direction, streams, frames per stream at most (user side))
    (44100, 48000, 2, PUSH, 23, 60000),
    (16000, 48000, 1, PULL, 29, 90000),     # two interpolators folded into the fractional stage's filter
    (48000, 44100, 2, PUSH, 11, 50000),
]


@pytest.mark.parametrize("src,dst,ch,direction,n,most", RAGGED)
def test_fast_mode_ragged_streams_with_a_reset(src, dst, ch, direction, n, most):
    """Every stream its own length and its own step pattern (steps of no frames,
    of a few hundred, of tens of thousands: the step's longest stream picks the
    kernel, so the staged and the split kernels hand a stream's history to
    each other), and one stream started over in the middle: each stream against
    an upstream converter of its own."""
    rng = np.random.default_rng(src + n)
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, direction, support.GOOD)
    assert status == 1
    config = drbamd.make_config(src, dst, ch, 256, 16, direction, support.GOOD, arithmetic=drbamd.FAST, streams=n)
    assert drbamd.plan_info(config).fast_path == 1
    inputs = [support.white_noise(ch, most, 300 + s) for s in range(n)]
    totals = [int(v) for v in rng.integers(most // 4, most, n)]
    totals[0] = 0
    runner = gpu_support.StreamRunner(config, inputs)
    left = list(totals)
    again, restarted, step_index = 3, False, 0
    while any(left):
        limit = (300, 5000, 40000)[step_index % 3]
        steps = [min(l, int(rng.integers(0, limit))) for l in left]
        runner.step(steps)
        left = [l - s for l, s in zip(left, steps)]
        step_index += 1
        if step_index == 4 and not restarted:
            # stream `again` starts over: what it made so far is dropped
            runner.batch.reset(again, 1)
            runner.read[again] = 0
            runner.outputs[again] = []
            left[again] = totals[again]
            restarted = True
    assert restarted
    for s in range(n):
        if direction == PUSH:
            want = support.oracle_push(plan, inputs[s][:, :totals[s]])
        else:
            want, _ = support.oracle_pull(plan, inputs[s], totals[s])
        got = runner.result(s)
        assert got.shape == want.shape, s
        if want.size:
            assert_close(got, want, "stream %d" % s)
    runner.close()


@pytest.mark.parametrize("src,dst,quality,kernel", [(176400, 192000, support.POOR, b"drb_fast_kernel"),
                                                    (176400, 192000, support.POOR, b"drb_split_kernel"),
                                                    (192000, 176400, support.BEST, b"drb_fast_kernel")])
def test_fast_mode_one_long_step_of_few_streams_on_the_staged_kernels(src, dst, quality, kernel, monkeypatch):
    """Few streams and one long step: a stream's tiles are dealt out to many
    workgroups, each of which builds its coefficients per tile; 176.4 <-> 192 kHz
    drifts 2.6 times as fast as 44.1 -> 48 kHz (coefficients built for a tile
    far from the one they were used on came to 6.9e-6 of error here).  A lone
    fractional stage (poor quality) runs the split kernel unless told otherwise:
    both are held to the same margin."""
    ch, frames = 2, 450000
    if quality == support.POOR and kernel == b"drb_fast_kernel":
        monkeypatch.setenv("DRB_AMD_FAST_LONE", "0")
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, quality)
    assert status == 1
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, arithmetic=drbamd.FAST, streams=2)
    info = drbamd.plan_info(config)
    assert info.fast_path == 1 and info.kernel_name == kernel
    inputs = [support.white_noise(ch, frames, 41 + s) for s in range(2)]
    runner = gpu_support.StreamRunner(config, inputs)
    runner.step([frames, frames - 4096])
    for s, total in enumerate((frames, frames - 4096)):
        want = support.oracle_push(plan, inputs[s][:, :total])
        got = runner.result(s)
        assert got.shape == want.shape
        assert assert_close(got, want, "stream %d" % s) < 9e-7
    runner.close()


@pytest.mark.parametrize("bad", [1.0e5, -7.0e4, float("inf"), float("nan")], ids=["1e5", "-7e4", "inf", "nan"])
def test_fast_mode_reports_samples_its_halves_cannot_carry(bad):
    """Upstream takes any float (drb-audio-converter.h:87-91); fast arithmetic
    carries a sample as two halves, which end at +-65504.  A sample beyond that
    (or an infinite one, or a NaN) in one stream of a batch: synchronize says so,
    once, the other streams are within the bound, and the batch goes on working.
    Exact arithmetic takes the same data and equals upstream's scalar build
    wherever that is a number (bit for bit), and is no number where it is none."""
    ch, frames, n = 2, 60000, 3
    status, plan = support.oracle_plan(44100, 48000, ch, 256, 16, PUSH, support.GOOD)
    assert status == 1
    inputs = [support.white_noise(ch, 2 * frames, 900 + s) for s in range(n)]
    inputs[1][1, 31234] = bad
    for arithmetic in (drbamd.FAST, drbamd.EXACT):
        config = drbamd.make_config(44100, 48000, ch, 256, 16, PUSH, support.GOOD, arithmetic=arithmetic, streams=n)
        batch = drbamd.Batch(config)
        source = torch.from_numpy(np.stack(inputs)).cuda()
        capacity = batch.query([2 * frames] * n)[0] + 4096
        target = torch.zeros((n, ch, capacity), device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        counts = (ctypes.c_int * n)()
        batch.process_dense(source.data_ptr(), ch * 2 * frames, 2 * frames, target.data_ptr(), ch * capacity, capacity,
                            frames, capacity, stream=stream, produced=counts)
        made = counts[0]
        if arithmetic == drbamd.FAST:
            with pytest.raises(drbamd.DrbError) as error:
                batch.synchronize(stream)
            assert error.value.status == "bad_argument" and "range" in str(error.value)
            batch.synchronize(stream)   # reported once
        else:
            batch.synchronize(stream)
        # a second step, of good samples only (the bad one is 28 766 frames back: out of every window)
        batch.process_dense(source.data_ptr() + 4 * frames, ch * 2 * frames, 2 * frames,
                            target.data_ptr() + 4 * made, ch * capacity, capacity,
                            frames, capacity - made, stream=stream, produced=counts)
        batch.synchronize(stream)
        made += counts[0]
        got = target.cpu().numpy()
        for s in range(n):
            want = support.oracle_push(plan, inputs[s])
            assert want.shape[1] == made
            numbers = np.isfinite(want)
            if arithmetic == drbamd.EXACT:
                assert np.array_equal(got[s, :, :made].view(np.uint32)[numbers], want.view(np.uint32)[numbers])
                assert not np.isfinite(got[s, :, :made][~numbers]).any()
            elif s != 1:
                assert numbers.all()
                assert_close(got[s, :, :made], want, "stream %d" % s)
            else:
                # the stream with the bad sample: right again once the sample has left every window
                tail = slice(frames * 48000 // 44100 + 2048, made)
                assert_close(got[s, :, tail], want[:, tail], "stream 1 behind the bad sample")
        batch.close()


@pytest.mark.parametrize("entry", ["regular", "static"])
def test_more_phases_than_coefficient_sets_fit_runs_in_groups_of_streams(entry, monkeypatch):
    """16 -> 48 kHz has its two interpolators folded into the one filter and no
    staged kernels to fall back to.  64 streams, every one at a phase of its
    own, with room for about twenty coefficient sets at a time (the limit
    lowered through the environment), or 96 streams in the static workspace,
    which holds 65 of this chain's sets: the step is not refused, it runs in
    groups of streams, every stream against an upstream converter of its own."""
    src, dst, ch, n = 16000, 48000, 1, 64 if entry == "regular" else 96
    if entry == "regular":
        monkeypatch.setenv("DRB_AMD_SET_BYTES_LIMIT", "1800000")
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, support.GOOD)
    assert status == 1
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, support.GOOD, arithmetic=drbamd.FAST, streams=n)
    info = drbamd.plan_info(config)
    assert info.fast_path == 1 and info.fast_launches == 1 and info.fast_collapsed == 2
    first = [1001 + 37 * s for s in range(n)]     # every stream its own length: phases apart from here on
    long = 3 * src
    inputs = [support.white_noise(ch, first[s] + long, 700 + s) for s in range(n)]
    runner = gpu_support.StreamRunner(config, inputs)
    runner.step(first)
    if entry == "regular":
        runner.step([long] * n)
        assert runner.batch.step_groups() > 3
        got = [runner.result(s) for s in range(n)]
    else:
        before = [runner.result(s) for s in range(n)]
        rest = np.stack([inputs[s][:, first[s]:] for s in range(n)])
        source = torch.from_numpy(rest).cuda()
        capacity = 3 * long + 4096
        target = torch.zeros((n, ch, capacity), device="cuda:0")
        counts = (ctypes.c_int * n)()
        stream = torch.cuda.current_stream().cuda_stream
        runner.batch.process_dense_static(source.data_ptr(), ch * long, long, target.data_ptr(), ch * capacity, capacity,
                                          long, capacity, stream=stream, produced=counts)
        runner.batch.synchronize(stream)
        assert runner.batch.step_groups() > 1
        out = target.cpu().numpy()
        got = [np.concatenate([before[s], out[s, :, :counts[s]]], axis=1) for s in range(n)]
    for s in range(n):
        want = support.oracle_push(plan, inputs[s])
        assert got[s].shape == want.shape, s
        assert_close(got[s], want, "stream %d" % s)
    runner.close()


WALKED = [
    # name, src, dst, quality, sample, layout, streams, workgroups
    ("headline planar float32", 44100, 48000, support.GOOD, "f32", "planar", 7, 3),
    ("int16 interleaved", 48000, 44100, support.FINE, "i16", "interleaved", 7, 3),
    ("cfg2 int16 interleaved", 48000, 44100, support.GOOD, "i16", "interleaved", 7, 3),
    ("int16 planar", 48000, 44100, support.GOOD, "i16", "planar", 5, 4),
    ("headline at best quality", 44100, 48000, support.BEST, "f32", "planar", 9, 2),
    ("float32 interleaved", 44100, 48000, support.FINE, "f32", "interleaved", 9, 2),
    ("fewer phases than a launch's sets", 88200, 96000, support.GOOD, "f32", "planar", 6, 5),
]


@pytest.mark.parametrize("case", WALKED, ids=[c[0] for c in WALKED])
def test_workgroups_that_go_from_stream_to_stream(case, monkeypatch):
    """Steps in which every stream has as many tiles run with fewer workgroups than
    streams and channels: each goes on from one stream to the next (here forced
    to 2 .. 5 workgroups, unevenly loaded; coefficient sets renewed every 2
    tiles, so that a workgroup's tiles of one stream fall into several
    stretches).  The streams are at phases of their own after a first step of
    uneven lengths; then come steps of 4096 frames, one of 3 frames, one long
    one, and one in which one stream gets more tiles than the others (which
    must run the ordinary way)."""
    name, src, dst, quality, sample, layout, n, workgroups = case
    ch = 2
    monkeypatch.setenv("DRB_AMD_UNIT_WALK", str(workgroups))
    monkeypatch.setenv("DRB_AMD_FAST_APART", "0")
    monkeypatch.setenv("DRB_AMD_FAST_REBUILD_TILES", "2")
    sample_type = {"f32": drbamd.F32, "i16": drbamd.I16}[sample]
    layout_type = drbamd.INTERLEAVED if layout == "interleaved" else drbamd.PLANAR
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, sample_type=sample_type, layout=layout_type,
                                arithmetic=drbamd.FAST, streams=n)
    info = drbamd.plan_info(config)
    assert info.fast_collapsed == 2
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, quality)
    assert status == 1
    first = [2 * (300 + 700 * s) + (s % 3 == 0) for s in range(n)]
    steps = [first, [4096] * n, [4096] * n, [3] * n, [4096] * n, [30000] * n, [4096 + (20000 if s == 2 else 0) for s in range(n)], [4096] * n]
    total = [sum(step[s] for step in steps) for s in range(n)]
    noise = [support.white_noise(ch, total[s], 900 + s) for s in range(n)]
    if sample == "i16":
        inputs = [np.round(x * 32767).astype(np.int16) for x in noise]
        floats = [support.oracle_convert_samples(x, "f32") for x in inputs]
    else:
        inputs, floats = noise, noise
    runner = gpu_support.StreamRunner(config, inputs)
    walked = 0
    for step in steps:
        produced = runner.step(step)
        tiles = set(-(-count // info.fast_tile_frames) for count in produced)
        expect = workgroups if len(tiles) == 1 and max(tiles) > 0 else 0
        assert runner.batch.step_walkers() == expect, (step[:3], runner.batch.step_walkers(), produced)
        walked += expect > 0
    assert walked >= 5
    for s in range(n):
        got = runner.result(s)
        want = support.oracle_push(plan, floats[s])
        assert got.shape == want.shape, (name, s, got.shape, want.shape)
        if sample == "i16":
            want_q = support.oracle_convert_samples(np.ascontiguousarray(want), "i16")
            assert np.abs(got.astype(np.int64) - want_q.astype(np.int64)).max() <= 1, (name, s)
        else:
            assert_close(got, want, "%s stream %d" % (name, s))
    runner.close()


DOWN_BY_THREE_QUARTERS = [
    # src, dst, quality, sample, layout: chains whose fractional stage goes 3 -> 4 (96 -> 128 kHz and the
    # like) in front of one to four decimators.  Planar float32: the buffers between a chain's launches
    # are that, and the split kernel takes a launch whose source and target have one format.
    (48000, 16000, support.FINE, "f32", "planar"),
    (48000, 16000, support.GOOD, "f32", "planar"),
    (48000, 16000, support.BEST, "f32", "planar"),
    (48000, 8000, support.GOOD, "f32", "planar"),       # (ran the staged kernels for this launch before)
    (96000, 16000, support.GOOD, "f32", "planar"),
    (96000, 8000, support.FINE, "f32", "planar"),
    (192000, 16000, support.GOOD, "f32", "planar"),
    (192000, 16000, support.BEST, "f32", "planar"),
]


@pytest.mark.parametrize("src,dst,quality,sample,layout", DOWN_BY_THREE_QUARTERS)
def test_fast_mode_segments_chosen_for_the_split_kernel(src, dst, quality, sample, layout):
    """48 / 96 / 192 kHz down to 8 / 16 kHz: the launch with the fractional stage has
    segments of 3 k source and 4 k target frames.  The fullest segment, 96 -> 128
    frames, has an even source length below what a padded plane takes; the planner
    goes on to 117 -> 156 (odd: a straight plane, ten row blocks), and the chain
    runs on the matrix cores.  Three streams of two seconds in calls of uneven
    length against the upstream converter's restatement."""
    ch = 2
    sample_type = {"f32": drbamd.F32, "i16": drbamd.I16}[sample]
    layout_type = drbamd.INTERLEAVED if layout == "interleaved" else drbamd.PLANAR
    config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, sample_type=sample_type, layout=layout_type,
                                arithmetic=drbamd.FAST, streams=3)
    info = drbamd.plan_info(config)
    assert info.fast_path == 1 and info.fast_collapsed == 2 and info.fast_why_not == b"", info.fast_why_not
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, PUSH, quality)
    assert status == 1
    lengths = [2 * src, int(1.3 * src) // 256 * 256, 2 * src]
    patterns = [[10 ** 9], [4096 * 7, 256, 4096 * 11], [src // 256 * 256, 4096]]
    noise = [support.white_noise(ch, lengths[s], 500 + s) for s in range(3)]
    if sample == "i16":
        inputs = [np.round(x * 32767).astype(np.int16) for x in noise]
        floats = [support.oracle_convert_samples(x, "f32") for x in inputs]
    else:
        inputs, floats = noise, noise
    runner = gpu_support.StreamRunner(config, inputs)
    schedules = [gpu_support.chunk_schedule(lengths[s], patterns[s]) for s in range(3)]
    for i in range(max(len(s) for s in schedules)):
        runner.step([s[i] if i < len(s) else 0 for s in schedules])
    for s in range(3):
        got = runner.result(s)
        want = support.oracle_push(plan, floats[s])
        assert got.shape == want.shape, (s, got.shape, want.shape)
        if sample == "i16":
            want_q = support.oracle_convert_samples(np.ascontiguousarray(want), "i16")
            assert np.abs(got.astype(np.int64) - want_q.astype(np.int64)).max() <= 1, s
        else:
            assert_close(got, want, "stream %d" % s)
    runner.close()


def test_plan_info_says_what_keeps_a_chain_from_the_matrix_cores():
    """A configuration that runs the exact kernels in fast arithmetic says why: which launch
    of its chain, and what that launch lacks."""
    cases = [
        (8000, 11025, support.GOOD, drbamd.F32, b"no tile geometry"),      # a period of 441 target frames
        (22050, 48000, support.POOR, drbamd.F32, b"no kernel image"),      # 147 -> 320 with one up-sampler
        (44100, 48000, support.GOOD, drbamd.I32, b"int32 samples"),
    ]
    for src, dst, quality, sample_type, text in cases:
        config = drbamd.make_config(src, dst, 2, 256, 16, PUSH, quality, sample_type=sample_type, arithmetic=drbamd.FAST, streams=2)
        info = drbamd.plan_info(config)
        assert info.fast_path == 0 and text in info.fast_why_not, (src, dst, info.fast_why_not)
    config = drbamd.make_config(44100, 48000, 2, 256, 16, PUSH, support.GOOD, arithmetic=drbamd.FAST, streams=2)
    assert drbamd.plan_info(config).fast_why_not == b""


FOLDED_FORMATS = [
    # src, dst, quality, sample, layout, direction: chains of interpolators folded into the one filter,
    # in the buffer formats other than planar float32
    (16000, 48000, support.GOOD, "i16", "interleaved", PUSH),
    (16000, 48000, support.GOOD, "i16", "interleaved", PULL),
    (8000, 48000, support.GOOD, "f32", "interleaved", PUSH),
    (8000, 48000, support.FINE, "f32", "interleaved", PULL),
    (16000, 96000, support.FINE, "i16", "planar", PUSH),
    (8000, 96000, support.POOR, "i16", "interleaved", PUSH),
]


@pytest.mark.parametrize("src,dst,quality,sample,layout,direction", FOLDED_FORMATS)
def test_fast_mode_folded_interpolators_in_every_buffer_format(src, dst, quality, sample, layout, direction):
    """8 / 16 kHz up to 48 / 96 kHz run as one launch, the interpolators folded into the
    fractional stage's filter, with tiles of 4 or 6 wave-wide loads; those kernel
    images exist for int16 and interleaved buffers as they do for planar float32.
    Three stereo streams of uneven length in calls of uneven length."""
    three_streams_against_the_oracle(src, dst, quality, sample, layout, direction, launches=1)


FIRST_LAUNCH_CONVERTS = [
    # src, dst, quality, sample, layout, direction, launches: decimating chains whose first launch (the one
    # with the fractional stage) reads the caller's format and hands planar float32 to the decimators behind it
    (48000, 16000, support.GOOD, "i16", "interleaved", PUSH, 3),
    (192000, 44100, support.BEST, "i16", "interleaved", PUSH, 3),    # (ran the staged kernels for that launch before)
    (96000, 22050, support.GOOD, "i16", "interleaved", PUSH, 3),     # (the same)
    (48000, 22050, support.GOOD, "f32", "interleaved", PUSH, 2),
    (88200, 48000, support.FINE, "i16", "planar", PUSH, 2),
    (96000, 44100, support.GOOD, "i16", "interleaved", PULL, 2),
    (176400, 48000, support.BEST, "f32", "interleaved", PUSH, 3),
    (48000, 11025, support.BEST, "i16", "interleaved", PULL, 3),
]


@pytest.mark.parametrize("src,dst,quality,sample,layout,direction,launches", FIRST_LAUNCH_CONVERTS)
def test_fast_mode_first_launch_reads_the_callers_format(src, dst, quality, sample, layout, direction, launches):
    """A chain of several launches keeps planar float32 between them.  The split kernel
    of its first launch loads int16 or interleaved samples as that format's kernel
    does and stores planar float32 (16-bit samples scaled back by a power of two);
    the last launch, a decimator, stores the caller's format."""
    three_streams_against_the_oracle(src, dst, quality, sample, layout, direction, launches=launches)


def three_streams_against_the_oracle(src, dst, quality, sample, layout, direction, launches, ch=2, collapsed=2):
    sample_type = {"f32": drbamd.F32, "i16": drbamd.I16}[sample]
    layout_type = drbamd.INTERLEAVED if layout == "interleaved" else drbamd.PLANAR
    config = drbamd.make_config(src, dst, ch, 256, 16, direction, quality, sample_type=sample_type, layout=layout_type,
                                arithmetic=drbamd.FAST, streams=3)
    info = drbamd.plan_info(config)
    assert info.fast_path == 1 and info.fast_collapsed == collapsed and info.fast_launches == launches, (info.fast_launches, info.fast_why_not)
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, direction, quality)
    assert status == 1
    user_rate = dst if direction == PULL else src
    lengths = [2 * user_rate // 256 * 256, int(1.3 * user_rate) // 256 * 256, 2 * user_rate // 256 * 256]
    patterns = [[10 ** 9], [4096 * 3, 256, 4096 * 5], [user_rate // 256 * 256, 4096]]
    need = max(lengths) if direction == PUSH else int(max(lengths) * src / dst) + 65536
    noise = [support.white_noise(ch, need, 600 + s) for s in range(3)]
    if sample == "i16":
        inputs = [np.round(x * 32767).astype(np.int16) for x in noise]
        floats = [support.oracle_convert_samples(x, "f32") for x in inputs]
    else:
        inputs, floats = noise, noise
    runner = gpu_support.StreamRunner(config, inputs)
    schedules = [gpu_support.chunk_schedule(lengths[s], patterns[s]) for s in range(3)]
    for i in range(max(len(s) for s in schedules)):
        runner.step([s[i] if i < len(s) else 0 for s in schedules])
    for s in range(3):
        got = runner.result(s)
        if direction == PUSH:
            want = support.oracle_push(plan, floats[s][:, :lengths[s]])
        else:
            want, _ = support.oracle_pull(plan, floats[s], lengths[s])
        assert got.shape == want.shape, (s, got.shape, want.shape)
        if sample == "i16":
            want_q = support.oracle_convert_samples(np.ascontiguousarray(want), "i16")
            assert np.abs(got.astype(np.int64) - want_q.astype(np.int64)).max() <= 1, s
        else:
            assert_close(got, want, "stream %d" % s)
    runner.close()


CONVERTED = [
    # src, dst, channels, quality, sample, layout, direction, launches, split form: the launches that planar
    # float32 gets, between one that turns the caller's buffers into planar float32 and one that turns the
    # result back (chains of 150 flop per target sample and more: cheaper ones stay on the exact kernels)
    (11025, 48000, 2, support.GOOD, "i16", "interleaved", PUSH, 5, 2),     # three interpolators, then the fractional stage's launch
    (22050, 96000, 3, support.FINE, "i16", "planar", PUSH, 5, 2),
    (48000, 11025, 5, support.GOOD, "f32", "interleaved", PULL, 3, 0),
    (176400, 44100, 6, support.GOOD, "i16", "interleaved", PUSH, 4, 0),    # two decimators
    (192000, 48000, 8, support.BEST, "f32", "interleaved", PUSH, 4, 0),
    (176400, 22050, 5, support.GOOD, "i16", "interleaved", PULL, 5, 0),    # pull, a slicer in front of three decimators
    (176400, 22050, 7, support.BEST, "i16", "interleaved", PULL, 5, 0),
]


@pytest.mark.parametrize("src,dst,ch,quality,sample,layout,direction,launches,collapsed", CONVERTED)
def test_fast_mode_buffer_formats_no_kernel_takes_go_through_planar_float32(src, dst, ch, quality, sample, layout, direction,
                                                                              launches, collapsed):
    """Interleaved buffers of more than two channels, and 16-bit or interleaved buffers at
    the ends of chains whose first or last launch keeps planar float32 on both sides,
    ran the exact kernels in fast arithmetic.  Now a launch that only converts (the
    exact path's sample conversions, frame for frame) stands in front of the chain
    and one behind it.  Three streams in calls of uneven length against the upstream
    converter's restatement; 16-bit results within one step."""
    three_streams_against_the_oracle(src, dst, quality, sample, layout, direction, launches=launches, ch=ch, collapsed=collapsed)


def test_cheap_chains_keep_the_exact_kernels_where_two_more_passes_would_not_pay():
    """One decimator or a few interpolators cost the exact kernels less than a pass over
    the samples costs the memory: on buffers no kernel of theirs takes, such chains
    stay there (measured 5 to 47 % faster than through planar float32), and the plan
    says so."""
    for src, dst, ch, quality in ((96000, 48000, 8, support.GOOD), (44100, 88200, 4, support.GOOD), (11025, 44100, 7, support.GOOD)):
        config = drbamd.make_config(src, dst, ch, 256, 16, PUSH, quality, sample_type=drbamd.I16, layout=drbamd.INTERLEAVED,
                                    arithmetic=drbamd.FAST, streams=2)
        info = drbamd.plan_info(config)
        assert info.fast_path == 0 and b"would not pay" in info.fast_why_not, (src, dst, info.fast_why_not)
