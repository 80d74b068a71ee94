# the exact kernel: its parity tests, cfg5 and the headline with exact arithmetic (1 s steps)
set -e
mkdir -p gpurun_out/ex
B="--no-cpu-baseline --no-host-fed --no-configs"
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_scale.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ex/tests.log 2>&1 || { tail -30 gpurun_out/ex/tests.log; exit 1; }
tail -2 gpurun_out/ex/tests.log
timeout -k 10 200 python3 bench.py --steps 10 --warmup 4 --workload cfg5_i32_best $B > gpurun_out/ex/cfg5.json 2> gpurun_out/ex/cfg5.err
timeout -k 10 200 python3 bench.py --steps 10 --warmup 4 --arithmetic exact --seconds 1 $B > gpurun_out/ex/head_exact.json 2> gpurun_out/ex/head_exact.err
timeout -k 10 200 python3 bench.py --steps 10 --warmup 4 --arithmetic exact --seconds 1 --workload cfg3_8ch $B > gpurun_out/ex/cfg3_exact.json 2> gpurun_out/ex/cfg3_exact.err
python3 - <<'PY'
import json
for f in ("cfg5", "head_exact", "cfg3_exact"):
    d = json.load(open("gpurun_out/ex/%s.json" % f))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
