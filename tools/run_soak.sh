# A long randomised parity run: every mode of tools/soak.py one after the other (about 20 minutes).
# usage: tools/run_soak.sh [first seed]     (logs under gpurun_out/soak/)
SEED=${1:-100}
mkdir -p gpurun_out/soak
status=0
timeout -k 10 400 python3 -u tools/soak.py --seconds 240 --seed $SEED > gpurun_out/soak/long_batch.log 2>&1 || status=1
timeout -k 10 300 python3 -u tools/soak.py --seconds 180 --seed $((SEED + 1)) --arithmetic fast > gpurun_out/soak/long_fast.log 2>&1 || status=1
timeout -k 10 300 python3 -u tools/soak.py --seconds 150 --seed $((SEED + 2)) --dense > gpurun_out/soak/long_dense.log 2>&1 || status=1
timeout -k 10 300 python3 -u tools/soak.py --seconds 150 --seed $((SEED + 3)) --api > gpurun_out/soak/long_api.log 2>&1 || status=1
timeout -k 10 300 python3 -u tools/soak.py --seconds 150 --seed $((SEED + 4)) --streams 3000 > gpurun_out/soak/long_many.log 2>&1 || status=1
timeout -k 10 400 python3 -u tools/soak.py --seconds 200 --seed $((SEED + 5)) --arithmetic fast --scale > gpurun_out/soak/long_scale.log 2>&1 || status=1
tail -q -n 1 gpurun_out/soak/long_*.log
exit $status
