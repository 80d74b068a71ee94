#!/bin/bash
# One GPU call: parity tests, the default bench line, a rocprofv3 kernel trace of
# the same command and the PMC passes for HBM traffic and issue counters.
# usage: tools/run_profile.sh [tag]   (results under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -2 $OUT/pytest_gpu.log
fi
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 15 --no-cpu-baseline --no-host-fed --no-configs"
mkdir -p $OUT/trace
cd /tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/trace/bench.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc --output-format csv -- $BENCH > $OUT/pmc_$name.log 2>&1 || { tail -20 $OUT/pmc_$name.log; exit 1; }
done
cd $ROOT
for d in $OUT/pmc_*/; do python3 tools/pmc_summary.py $d drb_; done > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
python3 - $OUT <<'PY' > $OUT/dispatches.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1] + "/trace/trace_kernel_trace.csv"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = {}
for r in rows:
    name = r["Kernel_Name"]
    if "drb_" not in name: continue
    short = name[name.find("drb_"):].split("(")[0]
    names.setdefault(short, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for short, times in names.items():
    last = times[-20:]
    print("%s: %d dispatches; ms in order: %s" % (short, len(times), " ".join("%.3f" % t for t in times)))
    print("%s: mean of the last %d (the timed steps): %.4f ms; of all: %.4f ms" % (short, len(last), sum(last) / len(last), sum(times) / len(times)))
PY
cat $OUT/dispatches.txt | cut -c1-400
