for a in 0 6 134 128 7; do
  echo "ablate $a: $(DRB_AMD_FAST_ABLATE=$a timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-host-fed --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["roofline"]["kernel_ms"])')"
done
