#!/bin/bash
# Kernel time of every bench workload with the staged and the split kernel.
set -o pipefail
mkdir -p gpurun_out
for w in headline headline_best headline_poor cfg2_f32 cfg2_i16 cfg2_i16_planar cfg4_mono_pull cfg5_i32_best; do
  for c in 0 1; do
    DRB_AMD_FAST_SPLIT=$c timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 2 --no-host-fed --no-cpu-baseline > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { echo "$w split=$c FAILED"; tail -3 gpurun_out/sweep.err; continue; }
    python3 -c '
import json,sys
d=json.load(open("gpurun_out/sweep.json")); r=d["roofline"]; print(sys.argv[1], "split="+sys.argv[2], "ms_per_step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], r["bound"], r["frac"], r.get("kernel"))' $w $c
  done
done
