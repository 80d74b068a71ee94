#!/usr/bin/env python3
"""Where the split kernel's waves spend their cycles (diagnostic build only).

Build `make -C audio-resampler_amd/csrc stamps`, then on the GPU box:
    DRB_AMD_LIB=audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so python3 tools/stamp_split.py
Prints the cycles per tile a filtering wave and a loader spend in each phase."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-resampler_amd"))

import torch

import drbamd

FILTER = ["bookkeeping", "barrier", "reads+products", "stores"]
LOADER = ["wait for loads", "convert", "issue", "barrier+idle"]


def main():
    streams, ch = 1024, 2
    cfg2 = os.environ.get("STAMP_CONFIG", "headline") == "cfg2"
    src_rate, dst_rate = (48000, 44100) if cfg2 else (44100, 48000)
    frames = int(os.environ.get("STAMP_FRAMES", str(10 * src_rate // 256 * 256)))
    config = drbamd.make_config(src_rate, dst_rate, ch, 256, 16, drbamd.PUSH, drbamd.GOOD,
                                sample_type=drbamd.I16 if cfg2 else drbamd.F32,
                                layout=drbamd.INTERLEAVED if cfg2 else drbamd.PLANAR,
                                arithmetic=drbamd.FAST, streams=streams)
    batch = drbamd.Batch(config)
    cap = batch.query([frames] * streams)[0] + 4096
    if cfg2:
        src = (torch.rand((streams, frames + 4096, ch), device="cuda") * 65000 - 32500).to(torch.int16)
        dst = torch.zeros((streams, cap, ch), device="cuda", dtype=torch.int16)
    else:
        src = torch.rand((streams, ch, frames + 4096), device="cuda") * 2 - 1
        dst = torch.zeros((streams, ch, cap), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    counters = (ctypes.c_ulonglong * 96)()
    drbamd.lib.drb_amd_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for step in range(3):
        batch.process_dense(src.data_ptr(), ch * (frames + 4096), frames + 4096, dst.data_ptr(), ch * cap, cap,
                            frames, cap, stream=stream)
        batch.synchronize(stream)
        drbamd.check(drbamd.lib.drb_amd_debug_counters(batch.handle, counters, 96))
    tile = 16 * 147 if cfg2 else 2560
    groups = 1 if cfg2 else ch   # interleaved stereo: one workgroup takes both channels through a tile
    tiles = streams * groups * ((frames * dst_rate // src_rate + tile - 1) // tile)
    blocks, loaders = 10, 6
    row = [counters[p] / blocks for p in range(4)]
    print("a filtering wave, %6.0f cycles per tile:  " % (sum(row) / tiles)
          + "  ".join("%s %.0f" % (FILTER[p], row[p] / tiles) for p in range(4)))
    row = [counters[16 + p] / loaders for p in range(4)]
    print("a loader,         %6.0f cycles per tile:  " % (sum(row) / tiles)
          + "  ".join("%s %.0f" % (LOADER[p], row[p] / tiles) for p in range(4))
          + ("  (it converts a third of every second tile)" if cfg2 else "  (it converts half of every third tile)"))
    if counters[9]:
        # (steps of a few tiles per stream run with as many workgroups as the chip holds: say how many)
        workgroups = int(os.environ.get("STAMP_WORKGROUPS", str(streams * groups)))
        life, real = counters[8] / workgroups, counters[9] / workgroups
        print("workgroup lifetime %.0f cycles = %.2f us: shader clock %.2f GHz" % (life, real / 100.0, life / real * 0.1))


if __name__ == "__main__":
    main()
