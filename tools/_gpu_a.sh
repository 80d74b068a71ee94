mkdir -p gpurun_out/r02x
O=gpurun_out/r02x
B="--steps 5 --warmup 2 --no-host-fed --no-cpu-baseline --no-configs"
timeout -k 10 200 python3 bench.py --workload headline --arithmetic exact --seconds 1 $B > $O/bench_headline_exact_1s.json 2> $O/err.txt || exit 1
timeout -k 10 200 python3 bench.py --workload cfg5_i32_best $B > $O/bench_cfg5.json 2>> $O/err.txt || exit 1
timeout -k 10 200 python3 bench.py --workload cfg3_8ch --arithmetic exact $B > $O/bench_cfg3_exact.json 2>> $O/err.txt || exit 1
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || exit 1
