# kernel time of the headline bench with phases of the collapsed kernel switched off
# (bits: 1 loader's loads, 2 products, 4 target stores, 8 coefficient reloads after the first, 16 drift update, 64 wide loads)
for a in ${ABLATIONS:-0 8 16 24 7 15 23 31}; do
  echo "ablate $a: $(DRB_AMD_FAST_ABLATE=$a timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-host-fed --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["roofline"]["kernel_ms"])')"
done
