#!/bin/bash
# kernel time of the exact kernel against the LDS a workgroup's two edge buffers may take
# (DRB_AMD_EXACT_LDS_BUDGET, bytes): larger tiles recompute less look-back, fewer workgroups fit a CU
set -e
mkdir -p gpurun_out/exact_budget
B="--no-cpu-baseline --no-host-fed --no-configs"
for kb in ${BUDGETS:-36 40 44 48 56 64 70 100 140}; do
  export DRB_AMD_EXACT_LDS_BUDGET=$((kb * 1024))
  a=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --steps 10 --warmup 3 $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['roofline']['kernel_ms'], d['config'].get('tile_frames'))")
  b=$(timeout -k 10 150 python3 bench.py --workload cfg5_i32_best --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['roofline']['kernel_ms'], d['config'].get('tile_frames'))")
  c=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --workload cfg3_8ch --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['roofline']['kernel_ms'], d['config'].get('tile_frames'))")
  echo "budget $kb KB: headline exact 1 s $a, cfg5 $b, cfg3 exact 1 s $c  (ms, tile)" | tee -a gpurun_out/exact_budget/times.txt
done
