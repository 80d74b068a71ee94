#!/bin/bash
# PMC passes for the int16 stereo configuration's split kernel (LDS use, conflicts, matrix pipe, waits).
# usage: tools/pmc_cfg2.sh <tag>
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --workload cfg2_i16 --steps 10 --warmup 5 --no-cpu-baseline --no-host-fed --no-configs"
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc --output-format csv -- $BENCH > $OUT/pmc_$name.log 2>&1 || { tail -20 $OUT/pmc_$name.log; exit 1; }
done
cd $ROOT
for d in $OUT/pmc_*/; do python3 tools/pmc_summary.py $d drb_; done > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
