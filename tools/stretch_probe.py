#!/usr/bin/env python3
"""How long a stretch of tiles one coefficient set of the split kernel can serve: worst
error against the oracle (and what is left of the bound 1e-6 + 1e-5 |y|) for stretches
of 64 (the default), 96, 128 and 192 tiles; 16 streams of 10 s, every stream pushed an odd
number of frames of its own first.  usage: python3 tools/stretch_probe.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "audio-resampler_amd"))

import support  # noqa: E402
import drbamd  # noqa: E402
import gpu_support  # noqa: E402


def probe(src, dst, quality, tiles, streams=16, ch=2):
    if tiles:
        os.environ["DRB_AMD_FAST_REBUILD_TILES"] = str(tiles)
    else:
        os.environ.pop("DRB_AMD_FAST_REBUILD_TILES", None)
    status, plan = support.oracle_plan(src, dst, ch, 256, 16, support.PUSH, quality)
    config = drbamd.make_config(src, dst, ch, 256, 16, support.PUSH, quality, arithmetic=drbamd.FAST, streams=streams)
    info = drbamd.plan_info(config)
    frames = 10 * src // 256 * 256
    rng = np.random.default_rng(9)
    first = [int(v) * 2 + 1 for v in rng.integers(50, 3000, streams)]
    inputs = [support.white_noise(ch, frames + first[s], 70 + s) for s in range(streams)]
    runner = gpu_support.StreamRunner(config, inputs)
    runner.step(first)
    runner.step([frames] * streams)
    worst_error, worst_excess = 0.0, -1.0
    for s in range(streams):
        want = support.oracle_push(plan, inputs[s])
        got = runner.result(s)
        error = np.abs(got.astype(np.float64) - want)
        worst_error = max(worst_error, float(error.max()))
        worst_excess = max(worst_excess, float((error - (1e-6 + 1e-5 * np.abs(want))).max()))
    runner.close()
    print("%d -> %d quality %d, %s, stretch %s tiles: worst error %.2e, over the bound by %.2e" % (
        src, dst, quality, info.kernel_name.decode(), tiles or "the library's choice (128, 64 or 32 by drift)", worst_error, worst_excess), flush=True)


for src, dst, quality in ((44100, 48000, 2), (48000, 44100, 2), (176400, 192000, 2), (44100, 48000, 3), (16000, 48000, 2)):
    for tiles in (0, 96, 128, 192):
        probe(src, dst, quality, tiles)
