set -o pipefail
mkdir -p gpurun_out/u16
for sec in 1 2 4; do
for walk in 0 256 0 256; do
DRB_AMD_UNIT_WALK=$walk timeout -k 10 200 python3 bench.py --seconds $sec --steps 40 --warmup 10 --no-host-fed --no-cpu-baseline --no-configs > gpurun_out/u16/s${sec}_$walk.json 2> gpurun_out/u16/s${sec}_$walk.err || { tail gpurun_out/u16/s${sec}_$walk.err; exit 1; }
python3 -c '
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("seconds", sys.argv[2], "walk", sys.argv[3], "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], r["frac"])' gpurun_out/u16/s${sec}_$walk.json $sec $walk
done
done
