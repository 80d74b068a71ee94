"""Pins the CPU restatement (oracle/) to the upstream converter.

Everything here runs without a GPU.  The golden fixtures were produced by the
upstream converter compiled from its own source (tests/golden/make_golden.py);
where that build is present (the build container) the oracle is additionally
checked against it live on fresh random cases.
"""
import numpy as np
import pytest

import golden_data
import support
from support import PUSH, PULL


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def plan_for(meta):
    status, plan = support.oracle_plan(meta["src"], meta["dst"], meta["channels"],
                                       meta["block_size"], meta["max_block_count"],
                                       meta["direction"], meta["quality"])
    assert status == 1
    return plan


META, DATA = golden_data.streams()


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_streams_bit_exact(name):
    meta = META[name]
    plan = plan_for(meta)
    for kind in meta["signals"]:
        x = DATA["%s.%s.in" % (name, kind)]
        want = DATA["%s.%s.out" % (name, kind)]
        if meta["direction"] == PUSH:
            got = support.oracle_push(plan, x)
        else:
            got, consumed = support.oracle_pull(plan, x, want.shape[1])
            assert consumed == meta["used"]
        assert got.shape == want.shape, (name, kind)
        assert np.array_equal(bits(got), bits(want)), (name, kind)


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_callback_counts(name):
    meta = META[name]
    plan = plan_for(meta)
    for sched in meta["callbacks"].values():
        got = support.oracle_callback_counts(plan, sched["chunks"])
        assert got.tolist() == sched["counts"], name


def stages_from_passes(passes, direction):
    """Golden pass lists are in planning order; pull chains run last to first."""
    order = passes if direction == PUSH else passes[::-1]
    out = []
    for tag in order:
        if tag in ("interleaver", "deinterleaver"):
            continue
        if tag.startswith("up"):
            out.append(("up2x", int(tag[2:])))
        elif tag.startswith("down"):
            out.append(("down2x", int(tag[4:])))
        elif tag.startswith("sinc"):
            out.append(("sinc", int(tag[4:-1]) + (100 if tag.endswith("c") else 0)))
        else:
            out.append(("slicer", None))
    return out


def slicer_block(passes, formats):
    at = passes.index("slicer")
    return max(formats[at][3], formats[at + 1][3])


def test_oracle_planner_matrix():
    checked = 0
    for row in golden_data.planner_rows():
        src, dst, ch, bs, mbc, direction, quality = row[:7]
        status, plan = support.oracle_plan(src, dst, ch, bs, mbc, direction, quality)
        if row[7] == "rejected":
            assert status == 0, row
            continue
        if row[7] == "unsupported":
            assert status == -1, row
            continue
        passes, formats, throttling = row[7], row[8], row[9]
        if status == -1:
            # the oracle declines plans the upstream driver cannot run
            assert mbc == 0 or throttling == 0, row
            continue
        assert status == 1, row
        want = stages_from_passes(passes, direction)
        got = plan.stages()
        assert [k for k, _ in got] == [k for k, _ in want], row
        for (kind, arg), (_, want_arg) in zip(got, want):
            if kind == "slicer":
                assert arg == slicer_block(passes, formats), row
            else:
                assert arg == want_arg, row
        assert plan.throttling == throttling, row
        checked += 1
    assert checked > 5000


def format_example(frames):
    lines = []
    for index in range(frames.shape[1]):
        lines.append("%3d" % index + "".join(", %8.3f" % v for v in frames[:, index]))
    return "\n".join(lines) + "\n"


def test_oracle_reproduces_example_push_output():
    slices = [13, 17, 4, 7, 5, 4, 21, 29, 300]
    x = np.stack([np.arange(500) + 100.0 * c for c in range(2)]).astype(np.float32)
    status, plan = support.oracle_plan(44100, 48000, 2, 4, 64, PUSH, support.GOOD)
    assert status == 1
    y = support.oracle_push(plan, x[:, :sum(slices)])
    assert format_example(y) == golden_data.example_text("example_push")


def test_oracle_reproduces_example_pull_output():
    slices = [13, 17, 4, 7, 5, 4, 21, 29, 300]
    status, plan = support.oracle_plan(44100, 48000, 2, 4, 256, PULL, support.GOOD)
    assert status == 1
    # the example's producer hands out frame + offset + 100 * channel
    x = np.stack([np.arange(4000) + 100.0 * c for c in range(2)]).astype(np.float32)
    y, _ = support.oracle_pull(plan, x, sum(slices))
    full = np.zeros((2, 500), np.float32)
    full[:, :sum(slices)] = y
    text = golden_data.example_text("example_pull")
    got = format_example(full).splitlines()
    want = text.splitlines()
    # frames past the requested 400 are uninitialised stack memory upstream
    assert got[:sum(slices)] == want[:sum(slices)]


def test_integer_sample_conventions():
    i16 = np.array([-32768, -1, 0, 1, 12345, 32767], np.int16)
    f = support.oracle_convert_samples(i16, "f32")
    assert np.array_equal(f, i16.astype(np.float32) / 32768.0)
    back = support.oracle_convert_samples(f, "i16")
    assert np.array_equal(back, i16)
    edge = np.array([1.0, -1.0, 0.99999, 2.0, -2.0, 0.5 / 32768, 1.5 / 32768, np.nan], np.float32)
    assert support.oracle_convert_samples(edge, "i16").tolist() == [32767, -32768, 32767, 32767, -32768, 0, 2, 0]
    assert support.oracle_convert_samples(edge, "i32").tolist()[:5] == [2147483647, -2147483648, 2147462144, 2147483647, -2147483648]
    i32 = np.array([-2**31, -12345678, 0, 1, 2**31 - 1], np.int32)
    f32 = support.oracle_convert_samples(i32, "f32")
    assert np.array_equal(f32, (i32.astype(np.float32) * np.float32(2.0 ** -31)))


@pytest.mark.skipif(not support.have_ref(), reason="upstream build only exists in the build container")
def test_oracle_against_upstream_build_live():
    rng = np.random.default_rng(7)
    ref = support.ref_harness()
    done = 0
    while done < 40:
        src, dst = (int(v) for v in rng.choice(support.RATES, 2))
        ch = int(rng.integers(1, 9))
        bs = int(rng.choice(support.BLOCK_SIZES[:5]))
        mbc = int(rng.integers(1, max(2, min(64, 8192 // bs))))
        direction = int(rng.integers(0, 2))
        quality = int(rng.integers(0, 4))
        status, plan = support.oracle_plan(src, dst, ch, bs, mbc, direction, quality)
        if status != 1 or not plan.consistent:
            continue
        chunks = [int(v) for v in rng.integers(0, 700, 6)]
        if direction == PUSH:
            x = support.white_noise(ch, sum(chunks), done)
            want, counts = ref.push(src, dst, ch, bs, mbc, quality, x, chunks)
            got = support.oracle_push(plan, x)
        else:
            need = support.oracle_lib().drb_oracle_pull_count(plan, sum(chunks))
            x = support.white_noise(ch, max(int(need), 1), done)
            want, counts, consumed = ref.pull(src, dst, ch, bs, mbc, quality, x, chunks)
            got, used = support.oracle_pull(plan, x, sum(chunks))
            assert used == consumed
        assert got.shape == want.shape
        assert np.array_equal(bits(got), bits(want)), (src, dst, ch, bs, mbc, direction, quality)
        assert support.oracle_callback_counts(plan, chunks).tolist() == counts.tolist()
        done += 1


@pytest.mark.skipif(not support.have_ref(), reason="upstream build only exists in the build container")
@pytest.mark.parametrize("case", [
    (8000, 11025, 2, 1, 64, PULL, support.POOR),     # a slicer of block 1 in the pull direction: one frame of silence first
    (44100, 48000, 2, 1, 2, PULL, support.POOR),
    (44100, 48000, 2, 1, 16, PULL, support.GOOD),
    (8000, 96000, 2, 64, 64, PUSH, support.FINE),    # eight stages: five interpolators, fractional stage, slicer, decimator
    (192000, 16000, 2, 256, 4, PUSH, support.FINE),  # eight stages the other way
    (8000, 96000, 2, 1, 1, PULL, support.GOOD),
])
def test_oracle_against_upstream_build_corner_chains(case):
    src, dst, ch, bs, mbc, direction, quality = case
    ref = support.ref_harness()
    status, plan = support.oracle_plan(src, dst, ch, bs, mbc, direction, quality)
    assert status == 1 and plan.consistent
    chunks = [300, 40, 0, 7000]
    if direction == PUSH:
        x = support.white_noise(ch, sum(chunks), 11)
        want, counts = ref.push(src, dst, ch, bs, mbc, quality, x, chunks)
        got = support.oracle_push(plan, x)
    else:
        need = support.oracle_lib().drb_oracle_pull_count(plan, sum(chunks))
        x = support.white_noise(ch, max(int(need), 1), 11)
        want, counts, consumed = ref.pull(src, dst, ch, bs, mbc, quality, x, chunks)
        got, used = support.oracle_pull(plan, x, sum(chunks))
        assert used == consumed
    assert got.shape == want.shape
    assert np.array_equal(bits(got), bits(want))
    assert support.oracle_callback_counts(plan, chunks).tolist() == counts.tolist()
