"""Worst error of the fast arithmetic against the CPU oracle as a function of how
often the coefficients are rebuilt from the exact phase (one workgroup walking
over all tiles of a 4 s stream), for the staged and for the split kernel.
Run on the GPU box:
    python3 tools/drift_probe.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "audio-resampler_amd")
import numpy as np
import support, gpu_support, drbamd
ch, frames = 2, 4 * 44100
x = support.white_noise(ch, frames, 21)
status, plan = support.oracle_plan(44100, 48000, ch, 256, 16, support.PUSH, support.GOOD)
want = support.oracle_push(plan, x).astype(np.float64)
for staged, rebuild in [(s, r) for s in ("1", "0") for r in (1, 4, 8, 16, 20, 32, 64)]:
    os.environ["DRB_AMD_FAST_SPLIT"] = "0" if staged == "1" else "1"
    os.environ["DRB_AMD_FAST_REBUILD_TILES"] = str(rebuild)
    os.environ["DRB_AMD_FAST_TILE_STRIDE"] = "1"
    config = drbamd.make_config(44100, 48000, ch, 256, 16, support.PUSH, support.GOOD, arithmetic=drbamd.FAST, streams=1)
    runner = gpu_support.StreamRunner(config, [x])
    runner.step([frames])
    got = runner.result(0).astype(np.float64)
    runner.close()
    err = np.abs(got - want)
    print("%s, rebuild every %2d tiles: worst %.3g  rms %.3g  worst excess over bound %.3g" % ("staged" if staged == "1" else "split", rebuild, err.max(), np.sqrt((err ** 2).mean()), (err - (1e-6 + 1e-5 * np.abs(want))).max()))
