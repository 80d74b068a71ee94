#!/usr/bin/env python3
"""Where the fast kernel's waves spend their cycles (diagnostic build only).

Build `make -C audio-resampler_amd/csrc stamps`, then on the GPU box:
    DRB_AMD_LIB=audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so python3 tools/stamp_fast.py
Prints, per wave of the workgroup, the share of cycles per phase."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-resampler_amd"))

import torch

import drbamd

PHASES = ["top", "load_x", "wait_b0", "store_x", "build", "up", "wait_b1", "sinc", "wait_b2", "down", "prefetch"]


def main():
    streams, ch, frames = 1024, 2, 44032
    config = drbamd.make_config(44100, 48000, ch, 256, 16, drbamd.PUSH, drbamd.GOOD,
                                arithmetic=drbamd.FAST, streams=streams)
    batch = drbamd.Batch(config)
    cap = batch.query([frames] * streams)[0] + 4096
    src = torch.rand((streams, ch, frames), device="cuda") * 2 - 1
    dst = torch.zeros((streams, ch, cap), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    counters = (ctypes.c_ulonglong * 96)()
    drbamd.lib.drb_amd_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for step in range(3):
        batch.process_dense(src.data_ptr(), ch * frames, frames, dst.data_ptr(), ch * cap, cap,
                            frames, cap, stream=stream)
        batch.synchronize(stream)
        drbamd.check(drbamd.lib.drb_amd_debug_counters(batch.handle, counters, 96))
    total_all = 0
    for wave in range(4):
        row = [counters[16 * wave + p] for p in range(len(PHASES))]
        total = sum(row)
        total_all += total
        print("wave %d  total %.1f Mcycles  " % (wave, total / 1e6)
              + "  ".join("%s %.1f%%" % (PHASES[p], 100.0 * row[p] / total) for p in range(len(PHASES))))
    print("cycles per tile iteration per wave: %.0f" % (total_all / 4 / (streams * ch * 20)))


if __name__ == "__main__":
    main()
