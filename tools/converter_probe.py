#!/usr/bin/env python3
"""How many one-stream batches (what a drop-in converter holds) fit without any being released.

usage: tools/converter_probe.py [most]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-resampler_amd"))

import torch

import drbamd


def main():
    most = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    torch.zeros(1, device="cuda")
    config = drbamd.make_config(44100, 48000, 2, 256, 16, drbamd.PUSH, drbamd.GOOD, streams=1)
    kept, start = [], time.perf_counter()
    try:
        while len(kept) < most:
            kept.append(drbamd.Batch(config))
            if len(kept) % 2000 == 0:
                free, total = torch.cuda.mem_get_info()
                print("%d batches, %.1f s, device memory in use %.2f GB" % (len(kept), time.perf_counter() - start, (total - free) / 1e9), flush=True)
    except drbamd.DrbError as error:
        print("creation %d failed: %s" % (len(kept) + 1, error))
    print("%d batches held, %.2f ms per creation" % (len(kept), (time.perf_counter() - start) / max(len(kept), 1) * 1e3))
    for batch in kept:
        batch.close()


if __name__ == "__main__":
    main()
