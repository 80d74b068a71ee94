#!/bin/bash
# Kernel trace of the headline fed in calls of 4096 frames (DRB_AMD_FAST_APART as given).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG:-short_trace}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o short --output-format csv -- python3 $ROOT/bench.py --call-frames 4096 --steps 2 --warmup 1 --no-host-fed --no-cpu-baseline --no-configs > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
cd $ROOT
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -12
python3 - $OUT <<'PY'
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2: len(rows) // 2 + 24]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f us  +%7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
PY
