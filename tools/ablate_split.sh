# kernel time of the headline bench with phases of the split kernel switched off
# (bits: 1 source loads, 2 conversion + LDS stores, 4 operand reads + products, 8 target stores)
for a in ${ABLATIONS:-0 1 2 3 4 8 12 7 15}; do
  echo "ablate $a: $(DRB_AMD_FAST_ABLATE=$a timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-host-fed --no-cpu-baseline --no-configs ${BENCH_ARGS} 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["roofline"]["kernel_ms"])')"
done
