#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the upstream converter.

Runs in the build container only: it needs oracle/_ref/libdrb_ref.so (the
upstream converter compiled from its own source by oracle/Makefile) and the two
upstream example programs, which it compiles from where they lie into a
temporary directory.  The upstream project ships no test vectors of its own,
so everything the parity tests are pinned to is produced here:

  planner.json.gz   resolved pass chains, per-edge formats, throttling, work
                    buffer length, converter and work-memory sizes for the
                    rate x rate x quality x direction matrix
  streams.npz       float32 inputs and the converter's float32 outputs (push
                    and pull, several chunk schedules folded into one output
                    because the output is chunk-invariant) + callback counts
  example_push.txt  stdout of the upstream example-push program
  example_pull.txt  stdout of the upstream example-pull program

Fixtures are data (inputs and expected outputs), never upstream source text.
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import support  # noqa: E402
from support import PUSH, PULL, POOR, FINE, GOOD, BEST, RATES  # noqa: E402

REF_DIR = "/root/reference"

# (name, src, dst, channels, block_size, max_block_count, direction, quality, input frames)
STREAM_CASES = [
    ("cfg1_push_good", 44100, 48000, 2, 4, 64, PUSH, GOOD, 6000),
    ("cfg1_pull_good", 44100, 48000, 2, 4, 64, PULL, GOOD, 6000),
    ("cfg1_push_poor", 44100, 48000, 2, 4, 64, PUSH, POOR, 4000),
    ("cfg1_push_fine", 44100, 48000, 2, 4, 64, PUSH, FINE, 4000),
    ("cfg1_push_best", 44100, 48000, 2, 4, 64, PUSH, BEST, 4000),
    ("cfg1_pull_best", 44100, 48000, 2, 4, 256, PULL, BEST, 4000),
    ("cfg1_push_good_bs1", 44100, 48000, 2, 1, 64, PUSH, GOOD, 3000),
    ("cfg2_push_good", 48000, 44100, 2, 256, 16, PUSH, GOOD, 6000),
    ("cfg3_push_good", 96000, 48000, 8, 256, 16, PUSH, GOOD, 6000),
    ("cfg4_pull_good", 16000, 48000, 1, 256, 16, PULL, GOOD, 4000),
    ("cfg4_push_good", 16000, 48000, 1, 256, 16, PUSH, GOOD, 3000),
    ("cfg5_push_best", 192000, 44100, 2, 256, 16, PUSH, BEST, 12000),
    ("cfg5_push_good", 192000, 44100, 2, 256, 16, PUSH, GOOD, 12000),
    ("up2_push_slicer", 48000, 96000, 2, 4, 64, PUSH, GOOD, 2000),
    ("up2_push_bs1", 48000, 96000, 3, 1, 64, PUSH, FINE, 2000),
    ("down2_pull_bs1", 96000, 48000, 2, 1, 64, PULL, GOOD, 4000),
    ("same_rate_reblock", 48000, 48000, 2, 16, 64, PUSH, GOOD, 2000),
    ("r8k_to_48k_push", 8000, 48000, 1, 4, 64, PUSH, GOOD, 1500),
    ("r48k_to_8k_pull", 48000, 8000, 1, 4, 64, PULL, GOOD, 12000),
    ("r11025_to_60k_pull_fine", 11025, 60000, 5, 16, 16, PULL, FINE, 1500),
    ("r240k_to_22050_push_poor", 240000, 22050, 2, 64, 8, PUSH, POOR, 20000),
    # a slicer of block 1 in the pull direction (its first frame is the silence it starts with)
    ("cfg1_pull_poor_bs1", 44100, 48000, 2, 1, 64, PULL, POOR, 3000),
    # the longest chains there are: five 2x stages on one side of the fractional stage, one on the other
    ("r8k_to_96k_push_fine", 8000, 96000, 2, 64, 64, PUSH, FINE, 1200),
    ("r192k_to_16k_push_fine", 192000, 16000, 2, 256, 4, PUSH, FINE, 30000),
]

SIGNALS = ["noise", "sweep"]
EXTRA_SIGNALS = {"cfg1_push_good": ["impulse", "dc", "ramp"], "cfg1_pull_good": ["impulse", "ramp"]}

SCHEDULES = {
    "example": [13, 17, 4, 7, 5, 4, 21, 29, 300],
    "ones": [1] * 40,
    "mixed": [1, 0, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610],
}


def make_signal(kind, channels, frames, rate, seed):
    if kind == "noise":
        return support.white_noise(channels, frames, seed)
    if kind == "sweep":
        return support.log_sweep(channels, frames, rate, 0.45 * rate)
    if kind == "impulse":
        x = np.zeros((channels, frames), np.float32)
        for c in range(channels):
            x[c, 100 + 7 * c] = 1.0
        return x
    if kind == "dc":
        return np.full((channels, frames), 0.75, np.float32)
    if kind == "ramp":  # the upstream examples' input: frame + 100 * channel
        return np.stack([np.arange(frames) + 100.0 * c for c in range(channels)]).astype(np.float32)
    raise ValueError(kind)


def chunking(total, step):
    chunks = [step] * (total // step)
    if total % step:
        chunks.append(total % step)
    return chunks


def run_case(ref, case, x):
    name, src, dst, ch, bs, mbc, direction, quality, frames = case
    if direction == PUSH:
        whole = ref.push(src, dst, ch, bs, mbc, quality, x, [frames])
        assert whole is not None, name
        y, _ = whole
        # the output stream does not depend on how the input is cut into calls
        for step in (1, 37, 500, 4096):
            other, _ = ref.push(src, dst, ch, bs, mbc, quality, x, chunking(frames, step))
            assert other.shape == y.shape and np.array_equal(other.view(np.uint32), y.view(np.uint32)), (name, step)
        return y, frames
    # pull: ask for as many target frames as the producer data can serve
    target = int(frames * dst / src)
    while True:
        y, _, consumed = ref.pull(src, dst, ch, bs, mbc, quality, x, [target])
        if consumed <= frames:
            break
        target -= 64
    for step in (1, 37, 500, 4096):
        other, _, _ = ref.pull(src, dst, ch, bs, mbc, quality, x, chunking(target, step))
        assert np.array_equal(other.view(np.uint32), y.view(np.uint32)), (name, step)
    return y, consumed


def make_streams(ref):
    arrays, meta = {}, {}
    for case in STREAM_CASES:
        name, src, dst, ch, bs, mbc, direction, quality, frames = case
        meta[name] = {"src": src, "dst": dst, "channels": ch, "block_size": bs,
                      "max_block_count": mbc, "direction": direction, "quality": quality,
                      "frames": frames, "signals": []}
        for kind in SIGNALS + EXTRA_SIGNALS.get(name, []):
            x = make_signal(kind, ch, frames, src, seed=len(name) * 131 + ch)
            y, used = run_case(ref, case, x)
            arrays["%s.%s.in" % (name, kind)] = x
            arrays["%s.%s.out" % (name, kind)] = y
            meta[name]["signals"].append(kind)
            meta[name]["used"] = int(used)
        # callback frame counts for irregular call schedules
        meta[name]["callbacks"] = {}
        for sched_name, schedule in SCHEDULES.items():
            x = make_signal("noise", ch, max(frames, sum(schedule)), src, 3)
            if direction == PUSH:
                _, counts = ref.push(src, dst, ch, bs, mbc, quality, x, schedule)
            else:
                big = support.white_noise(ch, 200000, 3)
                _, counts, _ = ref.pull(src, dst, ch, bs, mbc, quality, big, schedule)
            meta[name]["callbacks"][sched_name] = {"chunks": schedule, "counts": counts.tolist()}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "streams.npz"), **arrays)
    print("streams.npz:", len(arrays) - 1, "arrays")


def make_planner(ref):
    rows = []
    combos = [(2, 4, 64), (2, 1, 64), (2, 256, 16), (1, 256, 16), (8, 16, 32)]
    for ch, bs, mbc in combos:
        for src in RATES:
            for dst in RATES:
                for quality in (POOR, FINE, GOOD, BEST):
                    for direction in (PUSH, PULL):
                        plan = support.ref_plan(src, dst, ch, bs, mbc, direction, quality)
                        key = [src, dst, ch, bs, mbc, direction, quality]
                        if plan is None:
                            rows.append(key + ["rejected"])
                        elif plan == "unsupported":
                            rows.append(key + ["unsupported"])
                        else:
                            size, work = ref.sizes(src, dst, ch, bs, mbc, direction, quality)
                            rows.append(key + [plan["passes"], plan["formats"], plan["throttling"],
                                               plan["buffer_length"], size, work])
    # validation edge cases (everything here is either rejected or a plain plan)
    for args in [(44100, 48000, 2, 3, 64, PUSH, GOOD), (44100, 47999, 2, 4, 64, PUSH, GOOD),
                 (44100, 48000, 0, 4, 64, PUSH, GOOD), (44100, 48000, 9, 4, 64, PUSH, GOOD),
                 (44100, 48000, 2, 4, 8193, PUSH, GOOD), (44100, 48000, 2, 4096, 3, PUSH, GOOD),
                 (44100, 48000, 2, 4096, 2, PUSH, GOOD), (44100, 48000, 2, 4, -1, PULL, GOOD),
                 (0, 48000, 2, 4, 64, PULL, GOOD), (44100, 48000, 2, 1024, 8, PULL, BEST)]:
        plan = support.ref_plan(*args)
        if plan is None:
            rows.append(list(args) + ["rejected"])
        elif plan == "unsupported":
            rows.append(list(args) + ["unsupported"])
        else:
            size, work = ref.sizes(*args)
            rows.append(list(args) + [plan["passes"], plan["formats"], plan["throttling"],
                                      plan["buffer_length"], size, work])
    with gzip.open(os.path.join(HERE, "planner.json.gz"), "wt") as handle:
        json.dump(rows, handle, separators=(",", ":"))
    print("planner.json.gz:", len(rows), "rows")


def make_examples():
    with tempfile.TemporaryDirectory() as tmp:
        for name in ("example-push", "example-pull"):
            exe = os.path.join(tmp, name)
            subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-ffp-contract=off", "-I" + REF_DIR,
                                   "-o", exe, os.path.join(REF_DIR, name + ".c"),
                                   os.path.join(REF_DIR, "drb-audio-converter.c"), "-lm"])
            text = subprocess.check_output([exe]).decode()
            with open(os.path.join(HERE, name.replace("-", "_") + ".txt"), "w") as handle:
                handle.write(text)
            print(name, len(text.splitlines()), "lines")


if __name__ == "__main__":
    harness = support.ref_harness()
    make_examples()
    make_planner(harness)
    make_streams(harness)
