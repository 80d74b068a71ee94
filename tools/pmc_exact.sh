#!/bin/bash
# PMC passes for one bench workload (default: the headline in exact arithmetic).
# usage: tools/pmc_exact.sh <tag> [bench arguments]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-configs ${@:---arithmetic exact --seconds 1}"
for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_ADD_F32"
do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc --output-format csv -- $BENCH > $OUT/pmc_$name.log 2>&1 || { tail -20 $OUT/pmc_$name.log; exit 1; }
done
cd $ROOT
for d in $OUT/pmc_*/; do python3 tools/pmc_summary.py $d drb_cascade; done > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
