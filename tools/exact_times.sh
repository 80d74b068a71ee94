#!/bin/bash
# kernel times of the exact kernels for several builds of the library (tools/variant.sh):
# the headline in exact arithmetic at 1 s per step, and configuration 5 (int32, best)
for lib in ${LIBS:-tree}; do
  if [ $lib = tree ]; then unset DRB_AMD_LIB; else export DRB_AMD_LIB=audio-resampler_amd/lib/$lib/libdrb_audio_converter_amd.so; fi
  a=$(timeout 100 python3 bench.py --arithmetic exact --seconds 1 --steps 10 --warmup 3 --no-configs --no-cpu-baseline --no-host-fed 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  b=$(timeout 100 python3 bench.py --workload cfg5_i32_best --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-host-fed 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['roofline']['kernel_ms'])")
  echo "$lib: headline exact 1 s $a ms, cfg5 $b ms"
done
