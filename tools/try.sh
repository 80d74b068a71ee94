#!/bin/bash
# Quick GPU loop: fast-path parity tests, then kernel times of a few workloads
# (DRB_AMD_FAST_ABLATE values in ABLATIONS, workloads in WORKLOADS; LIBS="base tree"
# alternates builds kept under audio-resampler_amd/lib/<name>/ (tools/variant.sh) with the tree's).
set -o pipefail
OUT=gpurun_out/${TAG:-try}
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python3 -u -m pytest ${TESTS:-tests/test_gpu_fast.py} -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
fi
exec > >(tee $OUT/log.txt)
for w in ${WORKLOADS:-headline}; do
for a in ${ABLATIONS:-0}; do
for lib in ${LIBS:-tree}; do
if [ $lib = tree ]; then unset DRB_AMD_LIB; else export DRB_AMD_LIB=audio-resampler_amd/lib/$lib/libdrb_audio_converter_amd.so; fi
DRB_AMD_FAST_ABLATE=$a timeout -k 10 200 python3 bench.py --workload $w --steps ${STEPS:-20} --warmup 5 --no-host-fed --no-cpu-baseline --no-configs > $OUT/bench_${w}_$a.json 2> $OUT/bench_${w}_$a.err || { tail $OUT/bench_${w}_$a.err; exit 1; }
python3 -c '
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print(sys.argv[4], sys.argv[2], "ablate", sys.argv[3], "ms/step", d["ms_per_step"], "host", d["host_ms_per_step"], "kernel_ms", r["kernel_ms"], r["bound"], r["frac"], r["kernel"])' $OUT/bench_${w}_$a.json $w $a $lib
done
done
done
