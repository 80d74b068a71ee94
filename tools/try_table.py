#!/usr/bin/env python3
"""Kernel times of a tools/try.sh run as a table: python3 tools/try_table.py gpurun_out/<tag>/log.txt"""
import re, sys, collections
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+) ablate (\d+) ms/step (\S+) host (\S+) kernel_ms (\S+)", line)
    if m:
        rows.setdefault((m.group(1), m.group(2), m.group(3)), []).append((float(m.group(6)), float(m.group(4))))
for (lib, w, a), v in rows.items():
    ks = [x[0] for x in v]; ss = [x[1] for x in v]
    print("%-16s %-10s ablate %-3s kernel %s  min %.4f  step min %.4f" % (lib, w, a, " ".join("%.4f" % k for k in ks), min(ks), min(ss)))
