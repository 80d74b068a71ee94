#!/usr/bin/env python3
"""Copy a rocprofv3 --kernel-trace --stats run from gpurun_out/ into profiles/.

usage: tools/save_profile.py gpurun_out/prof_dir profiles/r01_name
Writes <dest>_kernel_stats.csv (kernel names cut to 120 characters) and
<dest>_bench.json (the benchmark's JSON line from bench.log, if present)."""
import csv
import glob
import os
import sys


def main(source, dest):
    stats = sorted(glob.glob(os.path.join(source, "**", "*kernel_stats.csv"), recursive=True))
    assert stats, "no kernel_stats.csv under " + source
    os.makedirs(os.path.dirname(dest) or ".", exist_ok=True)
    with open(stats[0]) as handle, open(dest + "_kernel_stats.csv", "w", newline="") as out:
        writer = csv.writer(out)
        for row in csv.reader(handle):
            row[0] = row[0][:120]
            writer.writerow(row)
    log = os.path.join(source, "bench.log")
    if os.path.exists(log):
        lines = [line for line in open(log) if line.startswith("{")]
        if lines:
            with open(dest + "_bench.json", "w") as out:
                out.write(lines[-1])
    print("saved", dest)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
