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
    which = os.environ.get("STAMP_CONFIG", "headline")
    cfg2, cfg4 = which == "cfg2", which == "cfg4"
    streams, ch = (4096, 1) if cfg4 else (1024, 2)
    src_rate, dst_rate = (48000, 44100) if cfg2 else (16000, 48000) if cfg4 else (44100, 48000)
    direction = drbamd.PULL if cfg4 else drbamd.PUSH
    user_rate = dst_rate if cfg4 else src_rate
    frames = int(os.environ.get("STAMP_FRAMES", str(10 * user_rate // 256 * 256)))   # the user's side: source (push) or target (pull)
    config = drbamd.make_config(src_rate, dst_rate, ch, 256, 16, direction, drbamd.GOOD,
                                sample_type=drbamd.I16 if cfg2 else drbamd.F32,
                                layout=drbamd.INTERLEAVED if cfg2 else drbamd.PLANAR,
                                arithmetic=drbamd.FAST, streams=streams)
    batch = drbamd.Batch(config)
    other = batch.query([frames] * streams)[0]
    source_frames, cap = (other + 4096, frames) if cfg4 else (frames + 4096, other + 4096)
    if cfg2:
        src = (torch.rand((streams, source_frames, ch), device="cuda") * 65000 - 32500).to(torch.int16)
        dst = torch.zeros((streams, cap, ch), device="cuda", dtype=torch.int16)
    else:
        src = torch.rand((streams, ch, source_frames), device="cuda") * 2 - 1
        dst = torch.zeros((streams, ch, cap), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    counters = (ctypes.c_ulonglong * 96)()
    drbamd.lib.drb_amd_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for step in range(3):
        if cfg4:
            batch.reset()
        batch.process_dense(src.data_ptr(), ch * source_frames, source_frames, dst.data_ptr(), ch * cap, cap,
                            frames, source_frames if cfg4 else cap, stream=stream)
        batch.synchronize(stream)
        drbamd.check(drbamd.lib.drb_amd_debug_counters(batch.handle, counters, 96))
    info = drbamd.plan_info(config)
    target_frames = frames if cfg4 else frames * dst_rate // src_rate
    tile = info.fast_tile_frames
    groups = 1 if cfg2 else ch   # interleaved stereo: one workgroup takes both channels through a tile
    tiles = streams * groups * ((target_frames + tile - 1) // tile)
    blocks, loaders = (tile // 16 + 15) // 16, 6   # a row block per 16 target frames of a segment
    row = [counters[p] / blocks for p in range(4)]
    print("a filtering wave, %6.0f cycles per tile:  " % (sum(row) / tiles)
          + "  ".join("%s %.0f" % (FILTER[p], row[p] / tiles) for p in range(4)))
    row = [counters[16 + p] / loaders for p in range(4)]
    print("a loader,         %6.0f cycles per tile:  " % (sum(row) / tiles)
          + "  ".join("%s %.0f" % (LOADER[p], row[p] / tiles) for p in range(4))
          + ("  (it converts a third of every second tile)" if cfg2 else "  (it converts half of every third tile)")
          + "   [%s, tile of %d target frames, %d row blocks]" % (which, tile, blocks))
    if counters[9]:
        # (steps of a few tiles per stream run with as many workgroups as the chip holds: say how many)
        workgroups = int(os.environ.get("STAMP_WORKGROUPS", str(streams * groups)))
        life, real = counters[8] / workgroups, counters[9] / workgroups
        print("workgroup lifetime %.0f cycles = %.2f us: shader clock %.2f GHz" % (life, real / 100.0, life / real * 0.1))


if __name__ == "__main__":
    main()
