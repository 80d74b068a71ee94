#!/bin/bash
# kernel time of the exact kernel against its tile (DRB_AMD_EXACT_TILE, target frames)
set -e
mkdir -p gpurun_out/exact_tiles
B="--no-cpu-baseline --no-host-fed --no-configs"
W=${WORKLOAD:-cfg5_i32_best}
for tile in ${TILES:-160 192 224 256 288 320 352 384 416}; do
  export DRB_AMD_EXACT_TILE=$tile
  b=$(timeout -k 10 150 python3 bench.py --workload $W $EXTRA --steps 5 --warmup 2 $B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['roofline']['kernel_ms'], d['config'].get('tile_frames'))")
  echo "$W $EXTRA tile $tile: $b" | tee -a gpurun_out/exact_tiles/times.txt
done
