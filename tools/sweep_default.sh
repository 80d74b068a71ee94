#!/bin/bash
# Every bench workload as the library picks its kernels by default; then the fast-path tests with the collapsed kernel forced.
for w in headline headline_best headline_poor cfg2_f32 cfg2_i16 cfg3_8ch cfg4_mono_pull cfg5_i32_best; do
    timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 2 --no-host-fed --no-cpu-baseline > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { echo "$w FAILED"; tail -3 gpurun_out/sweep.err; continue; }
    python3 -c '
import json,sys
d=json.load(open("gpurun_out/sweep.json")); r=d["roofline"]; print(sys.argv[1], "ms_per_step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], r["bound"], r["frac"], "issued", r.get("issued_mfma_frac"), r.get("kernel"))' $w
done
DRB_AMD_FAST_COLLAPSED=1 timeout -k 10 300 python3 -u -m pytest tests/test_gpu_fast.py -m gpu -q --timeout 120 --timeout-method thread 2>&1 | tail -3
