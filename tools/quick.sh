#!/bin/bash
# Quick GPU check: fast-path parity tests, then the headline bench without side measurements.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_fast.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_fast.log 2>&1 || { tail -30 gpurun_out/pytest_fast.log; exit 1; }
tail -1 gpurun_out/pytest_fast.log
for w in ${WORKLOADS:-headline}; do
timeout -k 10 200 python3 bench.py --workload $w --steps 20 --warmup 3 --no-host-fed --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail gpurun_out/bench_quick.err; exit 1; }
python3 -c '
import json,sys
d=json.load(open("gpurun_out/bench_quick.json")); r=d["roofline"]; print(sys.argv[1], "ms/step", d["ms_per_step"], "host", d["host_ms_per_step"], "kernel_ms", r["kernel_ms"], r["bound"], r["frac"], r["kernel"])' $w
done
