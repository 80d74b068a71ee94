set -o pipefail
mkdir -p gpurun_out/u01
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_fast.py -m gpu -x -q --timeout 120 --timeout-method thread ${PYTEST_ARGS:--k "stream_to_stream or many_tiles"} > gpurun_out/u01/pytest.log 2>&1 || { tail -40 gpurun_out/u01/pytest.log; exit 1; }
tail -1 gpurun_out/u01/pytest.log
for apart in ${APARTS:-default 0 1}; do
if [ $apart = default ]; then unset DRB_AMD_FAST_APART; else export DRB_AMD_FAST_APART=$apart; fi
DRB_AMD_HOST_PROFILE=1 timeout -k 10 200 python3 bench.py --call-frames 4096 --steps 3 --warmup 1 --no-host-fed --no-cpu-baseline --no-configs > gpurun_out/u01/short_$apart.json 2> gpurun_out/u01/short_$apart.err || { tail gpurun_out/u01/short_$apart.err; exit 1; }
python3 -c '
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "host", d["host_ms_per_step"], "kernel_ms", r["kernel_ms"], r["frac"], r["kernel"])' gpurun_out/u01/short_$apart.json $apart
grep "host us" gpurun_out/u01/short_$apart.err
done
if [ -n "$STAMPS" ]; then
DRB_AMD_FAST_APART=0 STAMP_FRAMES=4096 STAMP_WORKGROUPS=256 DRB_AMD_LIB=audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so timeout -k 10 200 python3 tools/stamp_split.py || exit 1
fi
