#!/usr/bin/env python3
"""Worst error of fast arithmetic against the CPU restatement: 10 s of noise
and of sweep, headline configuration, two streams (one call / two odd calls).
Prints |error| statistics and the worst excess over 1e-6 + 1e-5 |want|."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "audio-resampler_amd"), os.path.join(ROOT, "tests")]

import drbamd          # noqa: E402
import gpu_support     # noqa: E402
import support         # noqa: E402

src, dst, quality = 44100, 48000, support.GOOD
if len(sys.argv) > 3:
    src, dst, quality = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ch, frames = 2, 10 * src
status, plan = support.oracle_plan(src, dst, ch, 256, 16, support.PUSH, quality)
for signal in ("noise", "sweep", "quiet"):
    x = support.log_sweep(ch, frames, src, 0.45 * min(src, dst)) if signal == "sweep" else support.white_noise(ch, frames, 77)
    if signal == "quiet":
        x = (x * 1e-3).astype(np.float32)
    want = support.oracle_push(plan, x).astype(np.float64)
    config = drbamd.make_config(src, dst, ch, 256, 16, support.PUSH, quality, arithmetic=drbamd.FAST, streams=2)
    info = drbamd.plan_info(config)
    runner = gpu_support.StreamRunner(config, [x, x])
    schedules = [gpu_support.chunk_schedule(frames, p) for p in ([frames], [12345, frames - 12345])]
    for i in range(max(len(s) for s in schedules)):
        runner.step([s[i] if i < len(s) else 0 for s in schedules])
    for s in range(2):
        got = runner.result(s).astype(np.float64)
        err = np.abs(got - want)
        excess = err - (1e-6 + 1e-5 * np.abs(want))
        at = int(np.argmax(excess))
        print("%s stream %d (collapsed form %d): worst |err| %.3g, rms %.3g, worst excess %.3g at frame %d (want %.4g)"
              % (signal, s, info.fast_collapsed, err.max(), np.sqrt((err ** 2).mean()), excess.max(),
                 at % want.shape[1], want.reshape(-1)[at]))
    runner.close()
