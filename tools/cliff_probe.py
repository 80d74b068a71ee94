#!/usr/bin/env python3
"""Kernel time of one step of a rate pair in fast and in exact arithmetic (float32).

usage: tools/cliff_probe.py src dst quality [streams [seconds [channels [interleaved]]]]     quality: 0 poor .. 3 best"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-resampler_amd"))

import torch

import drbamd


def main():
    src, dst, quality = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    streams = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    seconds = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
    ch = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    interleaved = len(sys.argv) > 7 and sys.argv[7] not in ("0", "planar")
    frames = int(seconds * src) // 256 * 256
    source = torch.rand((streams, frames, ch) if interleaved else (streams, ch, frames), device="cuda") * 2 - 1
    stream = torch.cuda.current_stream().cuda_stream
    line = "%d -> %d Hz, quality %d, %d x %d-ch %s, %.1f s per step:" % (src, dst, quality, streams, ch, "interleaved" if interleaved else "planar", seconds)
    for name, arithmetic in (("fast", drbamd.FAST), ("exact", drbamd.EXACT)):
        config = drbamd.make_config(src, dst, ch, 256, 16, drbamd.PUSH, quality, arithmetic=arithmetic, streams=streams,
                                    layout=drbamd.INTERLEAVED if interleaved else drbamd.PLANAR)
        info = drbamd.plan_info(config)
        batch = drbamd.Batch(config)
        cap = batch.query([frames] * streams)[0] + 4096
        target = torch.zeros((streams, cap, ch) if interleaved else (streams, ch, cap), device="cuda")
        strides = ((ch * frames, 0), (ch * cap, 0)) if interleaved else ((ch * frames, frames), (ch * cap, cap))
        batch.time_kernels(True)
        times = []
        for _ in range(5):
            batch.process_dense(source.data_ptr(), *strides[0], target.data_ptr(), *strides[1], frames, cap, stream=stream)
            batch.synchronize(stream)
            times.append(batch.kernel_ms())
        batch.close()
        best = min(times[1:])
        made = streams * ch * (cap - 4096)
        line += "  %s %.3f ms (%.0f Gsamples/s, %s, %d launches)" % (name, best, made / best / 1e6, info.kernel_name.decode(), max(info.fast_launches, 1))
    print(line)


if __name__ == "__main__":
    main()
