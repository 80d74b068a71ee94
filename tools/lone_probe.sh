set -o pipefail
mkdir -p gpurun_out/u13
for lone in 0 1 0 1; do
DRB_AMD_FAST_LONE=$lone timeout -k 10 200 python3 bench.py --workload headline_poor --steps 20 --warmup 5 --no-host-fed --no-cpu-baseline --no-configs > gpurun_out/u13/poor_$lone.json 2> gpurun_out/u13/poor_$lone.err || { tail gpurun_out/u13/poor_$lone.err; exit 1; }
python3 -c '
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("lone", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], r["bound"], r["frac"], r["kernel"])' gpurun_out/u13/poor_$lone.json $lone
done
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_fast.py -m gpu -x -q --timeout 120 --timeout-method thread -k "poor or golden" > gpurun_out/u13/pytest.log 2>&1; tail -1 gpurun_out/u13/pytest.log
DRB_AMD_FAST_LONE=1 timeout -k 10 300 python3 -u -m pytest tests/test_gpu_fast.py tests/test_gpu_scale.py -m gpu -x -q --timeout 120 --timeout-method thread -k "poor or golden" > gpurun_out/u13/pytest_lone.log 2>&1; tail -1 gpurun_out/u13/pytest_lone.log
