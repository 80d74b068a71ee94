#!/usr/bin/env python3
"""Callback timestamps of the upstream converter (oracle/_ref, built from its own
source by oracle/Makefile) for a few conversions and chunk schedules ->
tests/golden/callback_stamps.json (doubles as hex strings, so nothing rounds).

Run here, where /root/reference exists:  python3 tests/golden/make_stamps.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import support  # noqa: E402

CASES = [
    # name, src, dst, channels, block, max blocks, direction, quality, chunks
    ("cfg1_push_good", 44100, 48000, 2, 4, 64, support.PUSH, support.GOOD, [13, 17, 4, 7, 5, 4, 21, 29, 300, 4096]),
    ("cfg1_pull_good", 44100, 48000, 2, 4, 256, support.PULL, support.GOOD, [13, 17, 4, 7, 5, 4, 21, 29, 300, 2048]),
    ("cfg1_push_poor", 44100, 48000, 2, 256, 16, support.PUSH, 0, [1000, 5000, 37]),
    ("cfg2_push_good", 48000, 44100, 2, 256, 16, support.PUSH, support.GOOD, [4096, 4096, 1234]),
    ("cfg3_push_good", 96000, 48000, 8, 256, 16, support.PUSH, support.GOOD, [4096, 512, 6000]),
    ("cfg4_pull_good", 16000, 48000, 1, 256, 16, support.PULL, support.GOOD, [4096, 3000, 512]),
    ("cfg5_push_best", 192000, 44100, 2, 256, 16, support.PUSH, 3, [8192, 20000]),
]


def main():
    harness = support.ref_harness()
    rows = {}
    for name, src, dst, ch, bs, mbc, direction, quality, chunks in CASES:
        x = support.white_noise(ch, 200000, 5)
        if direction == support.PUSH:
            _, counts = harness.push(src, dst, ch, bs, mbc, quality, x, chunks, timestamp=100.25)
        else:
            _, counts, _ = harness.pull(src, dst, ch, bs, mbc, quality, x, chunks, timestamp=100.25)
        rows[name] = {"args": [src, dst, ch, bs, mbc, direction, quality], "chunks": chunks, "timestamp": 100.25,
                      "counts": counts.tolist(), "stamps": [float(v).hex() for v in harness.stamps]}
        print(name, len(counts), harness.stamps[:3])
    with open(os.path.join(HERE, "callback_stamps.json"), "w") as out:
        json.dump(rows, out, indent=1)


if __name__ == "__main__":
    main()
