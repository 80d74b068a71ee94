"""Host-side logic of the product library against the golden planner matrix:
pass chains, edge formats, throttling, work-buffer length and converter sizes
must be exactly what the upstream planner resolves.  No GPU needed: these entry
points never touch the device."""
import ctypes

import pytest

import golden_data
import support

drbamd = pytest.importorskip("drbamd")


def config_of(row):
    src, dst, ch, bs, mbc, direction, quality = row[:7]
    return drbamd.make_config(src, dst, ch, bs, mbc, direction, quality)


def test_planner_matrix_matches_upstream():
    checked = rejected = 0
    for row in golden_data.planner_rows():
        src, dst, ch, bs, mbc, direction, quality = row[:7]
        described = drbamd.plan_describe(config_of(row))
        sizes = drbamd.converter_size(src, dst, ch, bs, mbc, direction, quality)
        if row[7] in ("rejected", "unsupported"):
            assert described is None and sizes is None, row
            rejected += 1
            continue
        passes, formats, throttling, buffer_length, size, work = row[7:13]
        assert described is not None, row
        got_passes, got_edges, got_throttling, got_length = described
        assert got_passes == passes, row
        assert got_edges == formats, row
        assert got_throttling == throttling, row
        assert got_length == buffer_length, row
        assert sizes == (64, size), row
        assert work == buffer_length * 2 * 4, row
        checked += 1
    assert checked > 5000 and rejected > 0


def test_plan_info_agrees_with_oracle_plan():
    for row in golden_data.planner_rows()[::7]:
        if row[7] in ("rejected", "unsupported"):
            continue
        src, dst, ch, bs, mbc, direction, quality = row[:7]
        status, plan = support.oracle_plan(src, dst, ch, bs, mbc, direction, quality)
        try:
            info = drbamd.plan_info(config_of(row))
        except drbamd.DrbError as error:
            assert error.status == "unsupported"
            assert status == -1 or not plan.consistent, row
            continue
        assert status == 1 and plan.consistent, row
        assert info.stages() == plan.stages(), row
        assert info.throttling == plan.throttling, row
        assert info.history_frames >= 4 and info.tile_frames >= 16


def test_flops_and_bytes_per_output_sample():
    # work per target sample of the cascade as written, 2 x MACs
    cases = {
        (44100, 48000, 2, support.GOOD, drbamd.F32): (424, 7.675),
        (44100, 48000, 2, support.POOR, drbamd.F32): (68, 7.675),
        (44100, 48000, 2, support.FINE, drbamd.F32): (456, 7.675),
        (44100, 48000, 2, support.BEST, drbamd.F32): (774, 7.675),
        (48000, 44100, 2, support.GOOD, drbamd.I16): (438, 4.177),
        (96000, 48000, 8, support.GOOD, drbamd.F32): (82, 12.0),
        (192000, 44100, 2, support.BEST, drbamd.I32): (3296, 21.415),
    }
    for (src, dst, ch, quality, sample), (flops, nbytes) in cases.items():
        config = drbamd.make_config(src, dst, ch, 256, 16, support.PUSH, quality, sample_type=sample)
        info = drbamd.plan_info(config)
        assert abs(info.flops_per_output_sample - flops) <= 0.01 * flops + 1, (src, dst, quality)
        assert abs(info.bytes_per_output_sample - nbytes) < 0.01


def test_library_exports_every_declared_symbol():
    names = ["drb_audio_converter_alignment_and_size", "drb_audio_converter_construct",
             "drb_audio_converter_work_memory_alignment_and_size", "drb_audio_converter_process",
             "drb_amd_converter_release", "drb_amd_batch_create", "drb_amd_batch_destroy",
             "drb_amd_batch_reset", "drb_amd_batch_query", "drb_amd_batch_process",
             "drb_amd_batch_process_dense", "drb_amd_batch_process_dense_static", "drb_amd_batch_process_host_dense", "drb_amd_batch_synchronize",
             "drb_amd_plan_info",
             "drb_amd_batch_time_kernels", "drb_amd_batch_kernel_ms", "drb_amd_batch_step_ms",
             "drb_amd_plan_describe", "drb_amd_last_error",
             "drb_amd_allocation_calls", "drb_amd_batch_step_groups", "drb_amd_batch_step_walkers", "drb_amd_host_times"]
    import re
    header = open(support.ROOT + "/include/drb_audio_converter_amd.h").read()
    declared = set(re.findall(r"\b(drb_a\w+)\s*\n?\s*\(", header))
    declared -= {"drb_audio_converter_version"}
    assert declared == set(names), declared ^ set(names)
    for name in names:
        assert getattr(drbamd.lib, name) is not None
    harness = ctypes.CDLL(drbamd.HARNESS_PATH)
    assert harness.harness_push and harness.harness_pull and harness.harness_sizes


def test_which_kernels_a_configuration_gets():
    """What drb_amd_plan_info reports about the matrix-core path: the benchmark
    configurations, chains of interpolators folded into one launch, the cases
    that run the exact kernels (host logic only: no device is touched)."""
    def info(src, dst, ch, direction, quality, **kw):
        return drbamd.plan_info(drbamd.make_config(src, dst, ch, 256, 16, direction, quality, **kw))

    fast = dict(arithmetic=drbamd.FAST, streams=4)

    headline = info(44100, 48000, 2, support.PUSH, support.GOOD, **fast)
    assert (headline.fast_path, headline.fast_launches, headline.fast_collapsed) == (1, 1, 2)
    assert headline.kernel_name == b"drb_split_kernel" and headline.fast_tile_frames == 16 * 160
    assert headline.fast_issued_half_flops_per_output_sample == headline.fast_issued_flops_per_output_sample > 0

    # one stage: nothing to fold, and still the split kernel (the faster one for it as well)
    poor = info(44100, 48000, 2, support.PUSH, support.POOR, **fast)
    assert (poor.fast_path, poor.fast_collapsed) == (1, 2) and poor.kernel_name == b"drb_split_kernel"

    # a lone half-band decimator
    cfg3 = info(96000, 48000, 8, support.PUSH, support.GOOD, **fast)
    assert cfg3.fast_path == 1 and cfg3.kernel_name == b"drb_decimate_kernel"

    # interpolators in front of the fractional stage fold into its launch (4 : 3 behind them) ...
    for src, dst in ((16000, 48000), (8000, 48000), (16000, 96000)):
        folded = info(src, dst, 1, support.PULL, support.GOOD, **fast)
        assert (folded.fast_path, folded.fast_launches, folded.fast_collapsed) == (1, 1, 2), (src, dst)
    # ... and stay launches of their own where no segment of whole periods fits (147 : 160 behind three of them)
    apart = info(22050, 96000, 2, support.PUSH, support.FINE, **fast)
    assert apart.fast_path == 1 and apart.fast_launches == 3

    # int32 buffers and exact arithmetic: the exact kernels, whatever was asked for
    assert info(192000, 44100, 2, support.PUSH, support.BEST, sample_type=drbamd.I32, **fast).fast_path == 0
    exact = info(44100, 48000, 2, support.PUSH, support.GOOD, streams=4)
    assert exact.fast_path == 0 and exact.kernel_name == b"drb_cascade_kernel"

    # a ratio whose period is longer than a segment may be: exact
    assert info(8000, 11025, 2, support.PUSH, support.GOOD, **fast).fast_path == 0
