# usage: VAR=NAME VALUES="a b c" bash tools/sweep_env.sh  -- headline kernel time per value of an environment variable
for v in $VALUES; do
  echo "$VAR=$v: $(env $VAR=$v timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-host-fed --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["ms_per_step"])')"
done
