#!/usr/bin/env python3
"""Mean value per dispatch of every counter in rocprofv3 --pmc output, per kernel.

usage: pmc_summary.py <directory written by rocprofv3 -d> [kernel name substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    wanted = sys.argv[2] if len(sys.argv) > 2 else ""
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(set))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                name = row.get("Kernel_Name", "")
                if wanted not in name:
                    continue
                short = name.split("(")[0][:60]
                counter = row["Counter_Name"]
                sums[short][counter] += float(row["Counter_Value"])
                counts[short][counter].add(row.get("Dispatch_Id", "0"))
    for kernel, counters in sums.items():
        print(kernel)
        for counter, total in sorted(counters.items()):
            dispatches = max(1, len(counts[kernel][counter]))
            print("  %-32s %16.0f per dispatch (%d dispatches)" % (counter, total / dispatches, dispatches))


if __name__ == "__main__":
    main()
