# the headline and configuration 2, library A (an older build under audio-resampler_amd/lib/ab) against B (the tree's), alternating
set -e
mkdir -p gpurun_out/ab
B="--no-cpu-baseline --no-host-fed --no-configs"
A=audio-resampler_amd/lib/ab/libdrb_audio_converter_amd.so
for round in 1 2 3; do
  DRB_AMD_LIB=$A timeout -k 10 200 python3 bench.py --steps 30 --warmup 15 $B > gpurun_out/ab/a_$round.json 2> gpurun_out/ab/a_$round.err
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 15 $B > gpurun_out/ab/b_$round.json 2> gpurun_out/ab/b_$round.err
done
DRB_AMD_LIB=$A timeout -k 10 200 python3 bench.py --steps 30 --warmup 15 --workload cfg2_i16 $B > gpurun_out/ab/a_cfg2.json 2> gpurun_out/ab/a_cfg2.err
timeout -k 10 200 python3 bench.py --steps 30 --warmup 15 --workload cfg2_i16 $B > gpurun_out/ab/b_cfg2.json 2> gpurun_out/ab/b_cfg2.err
