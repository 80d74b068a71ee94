#!/bin/bash
# Extra PMC passes for the headline kernel: vector-memory issue, LDS, L1 stalls, instruction cache.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-deep}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-fed"
i=0
for pass in "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
            "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TD_TD_BUSY TD_TC_STALL" \
            "TCP_GATE_EN1 TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCP_TA_DATA_STALL_CYCLES" \
            "SQC_ICACHE_MISSES SQC_ICACHE_REQ"
do
    i=$((i+1))
    timeout -s KILL 90 rocprofv3 --pmc $pass -d $OUT/p$i -o pmc --output-format csv -- $BENCH > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
cd $ROOT
for d in $OUT/p*/; do python3 tools/pmc_summary.py $d collapsed_plain; done | tee $OUT/summary.txt
