mkdir -p gpurun_out/r03u
O=gpurun_out/r03u
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q -s --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt; grep "excess" $O/tests.txt
timeout -k 10 300 python3 bench.py --steps 30 --warmup 20 --no-cpu-baseline --no-host-fed > $O/bench.json 2> $O/err.txt || exit 1
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
for c in d.get('configs',[]): r=c['roofline']; print(c.get('name'), c.get('value'), c.get('ms_per_step'), r['kernel'], r['kernel_ms'], r['bound'], r['frac'], r['device_step_ms'])"
