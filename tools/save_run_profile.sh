#!/bin/bash
# Copies what tools/run_profile.sh left under gpurun_out/<tag>/ into profiles/<name>_*.
# usage: tools/save_run_profile.sh <tag> <name>     e.g. r05a r02_headline_split
set -e
TAG=$1; NAME=$2
python3 tools/save_profile.py gpurun_out/$TAG/trace profiles/$NAME
cp gpurun_out/$TAG/dispatches.txt profiles/${NAME}_dispatches.txt
cp gpurun_out/$TAG/bench.json profiles/${NAME%%_*}_bench_default_with_configs.json
python3 - gpurun_out/$TAG/pmc_summary.txt profiles/${NAME}_pmc.txt <<'PY'
import re, sys
text = open(sys.argv[1]).read()
def per_dispatch(counter):
    # the last group of every block is the split kernel's
    found = re.findall(counter + r"\s+(\d+) per dispatch", text)
    return int(found[-1])
reads, writes = per_dispatch("TCC_EA0_RDREQ_sum"), per_dispatch("TCC_EA0_WRREQ_sum")
read_gb, written_gb = reads * 64 * 2 / 1e9, writes * 64 / 1e9
header = """rocprofv3 --pmc passes of `python3 bench.py --steps 20 --warmup 15 --no-cpu-baseline --no-host-fed --no-configs` (tools/run_profile.sh),
one pass per block of counters.  In every block the groups are: drb_table_kernel (44 dispatches); an unnamed group of 88 dispatches =
drb_collapse_build_kernel + drb_history_kernel together; and "void" (44 dispatches) = drb_split_kernel<4, 10, 0, false, false, false>, the
headline's filter kernel.
HBM traffic per launch of the split kernel: TCC_EA0_RDREQ_sum %s x 64 B x 2 (gfx950 tallies the 128-B requests of
16-B-per-lane streaming reads at 64 B) = %.3f GB read; TCC_EA0_WRREQ_sum %s x 64 B = %.3f GB written;
%.3f GB = %.3f x the algorithmic 7.55 GB.

""" % (format(reads, ","), read_gb, format(writes, ","), written_gb, read_gb + written_gb, (read_gb + written_gb) / 7.5497)
open(sys.argv[2], "w").write(header + text)
PY
