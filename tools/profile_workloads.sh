#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of bench workloads.
# usage: tools/profile_workloads.sh <tag> <workload> ...   (results under gpurun_out/<tag>/<workload>/)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for w in "$@"; do
    OUT=$ROOT/gpurun_out/$TAG/$w
    mkdir -p $OUT
    timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python3 $ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-host-fed --no-configs > $OUT/bench.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
    echo "== $w"; find $OUT -name "*kernel_stats.csv" | xargs -r cut -d, -f1-4 | cut -c1-150
done
