set -o pipefail
mkdir -p gpurun_out/f01
timeout -k 10 1100 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/f01/pytest.log 2>&1 || { tail -40 gpurun_out/f01/pytest.log; exit 1; }
tail -1 gpurun_out/f01/pytest.log
