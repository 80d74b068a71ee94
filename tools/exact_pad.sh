#!/bin/bash
# kernel time of the exact kernel against LDS asked for beyond its need (DRB_AMD_EXACT_LDS_PAD):
# the same work at fewer workgroups per CU
set -e
mkdir -p gpurun_out/exact_pad
B="--no-cpu-baseline --no-host-fed --no-configs"
for pad in ${PADS:-0 4096 8192 10240 11264 12288 16384 20480 24576 32768 40960 65536}; do
  export DRB_AMD_EXACT_LDS_PAD=$pad
  a=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --steps 10 --warmup 3 $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['roofline']['kernel_ms'])")
  c=$(timeout -k 10 150 python3 bench.py --arithmetic exact --seconds 1 --workload cfg3_8ch --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['roofline']['kernel_ms'])")
  echo "pad $pad: headline exact 1 s $a ms, cfg3 exact 1 s $c ms" | tee -a gpurun_out/exact_pad/times.txt
done
