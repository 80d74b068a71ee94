#!/bin/bash
# kernel time of the exact kernel with phases switched off (DRB_AMD_EXACT_ABLATE, timing only):
# where the time of configuration 5 and of the headline in exact arithmetic goes
set -e
mkdir -p gpurun_out/exact_ablate
B="--no-cpu-baseline --no-host-fed --no-configs"
for mask in 0 1 2 4 6 7 8 16 24 31 0; do
  export DRB_AMD_EXACT_ABLATE=$mask
  a=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --steps 10 --warmup 3 $B 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  b=$(timeout -k 10 150 python3 bench.py --workload cfg5_i32_best --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  echo "ablate=$mask: headline exact 1 s $a ms, cfg5 $b ms" | tee -a gpurun_out/exact_ablate/times.txt
done
