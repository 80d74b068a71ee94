#!/bin/bash
# Collapsed kernel only: bench, cycle stamps, then its parity tests.
set -o pipefail
mkdir -p gpurun_out
DRB_AMD_FAST_COLLAPSED=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-host-fed --no-cpu-baseline > gpurun_out/bench_collapsed.json 2> gpurun_out/bench_collapsed.err || { cat gpurun_out/bench_collapsed.err; exit 1; }
python3 -c '
import json
d=json.load(open("gpurun_out/bench_collapsed.json")); print("collapsed", d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])'
DRB_AMD_FAST_COLLAPSED=1 DRB_AMD_LIB=audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so timeout -k 10 200 python3 tools/stamp_collapsed.py > gpurun_out/stamps_collapsed.log 2>&1; cat gpurun_out/stamps_collapsed.log
DRB_AMD_FAST_COLLAPSED=1 timeout -k 10 300 python3 -u -m pytest tests/test_gpu_fast.py -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/pytest_collapsed.log 2>&1; tail -4 gpurun_out/pytest_collapsed.log
