for a in 0 1 2 4 8 14 15 7 11 13; do
  DRB_AMD_FAST_ABLATE=$a timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate',$a,'kernel_ms',d['roofline']['kernel_ms'])"
done
