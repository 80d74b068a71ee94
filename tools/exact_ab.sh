#!/bin/bash
# The exact kernel with a switch of its own in the environment on and off (VAR=name of the
# switch, default DRB_AMD_SINC_SHARE), on one box: parity tests first, then kernel times.
set -e
VAR=${VAR:-DRB_AMD_SINC_SHARE}
mkdir -p gpurun_out/exact_ab
B="--no-cpu-baseline --no-host-fed --no-configs"
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_scale.py tests/test_gpu_soak.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/exact_ab/tests.log 2>&1 || { tail -40 gpurun_out/exact_ab/tests.log; exit 1; }
tail -2 gpurun_out/exact_ab/tests.log
fi
for on in 0 1 0 1; do
  export $VAR=$on
  a=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --steps 10 --warmup 3 $B 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  b=$(timeout -k 10 150 python3 bench.py --workload cfg5_i32_best --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  c=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --workload cfg3_8ch --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  echo "$VAR=$on: headline exact 1 s $a ms, cfg5 $b ms, cfg3 exact 1 s $c ms" | tee -a gpurun_out/exact_ab/times.txt
done
