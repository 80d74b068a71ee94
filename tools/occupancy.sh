#!/bin/bash
# Wave lifetime and busy cycles of the headline kernel (one PMC pass), plus cycle stamps.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/occ
mkdir -p $OUT
cd $ROOT
DRB_AMD_LIB=audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so timeout -k 10 200 python3 tools/stamp_split.py || exit 1
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-fed > $OUT/pmc.log 2>&1 || { tail $OUT/pmc.log; exit 1; }
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc collapsed
grep -o '"kernel_ms": [0-9.]*' $OUT/pmc.log
