set -o pipefail
mkdir -p gpurun_out/u17
for walk in -1 0 -1 0; do
DRB_AMD_UNIT_WALK=$walk timeout -k 10 200 python3 bench.py --seconds 1 --phases-apart --steps 40 --warmup 10 --no-host-fed --no-cpu-baseline --no-configs > gpurun_out/u17/a_$walk.json 2> gpurun_out/u17/a_$walk.err || { tail gpurun_out/u17/a_$walk.err; exit 1; }
python3 -c '
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("phases apart, 1 s steps, walk", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], r["frac"])' gpurun_out/u17/a_$walk.json $walk
done
