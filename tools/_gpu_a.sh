mkdir -p gpurun_out/r02s
O=gpurun_out/r02s
B="--steps 30 --warmup 3 --no-host-fed --no-cpu-baseline --no-configs"
DRB_AMD_LIB=$PWD/audio-resampler_amd/lib/libdrb_head.so timeout -k 10 150 python3 bench.py --workload headline $B > $O/bench_head_1.json 2> $O/err.txt || exit 1
timeout -k 10 150 python3 bench.py --workload headline $B > $O/bench_now_1.json 2>> $O/err.txt || exit 1
for w in cfg2_i16 cfg2_f32 cfg2_i16_planar; do timeout -k 10 150 python3 bench.py --workload $w $B > $O/bench_$w.json 2>> $O/err.txt || exit 1; done
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_fast.py tests/test_gpu_scale.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.txt 2>&1 || exit 1
STAMP_CONFIG=cfg2 DRB_AMD_LIB=$PWD/audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so timeout -k 10 120 python3 tools/stamp_split.py > $O/stamps_cfg2.txt 2>&1 || exit 1
DRB_AMD_LIB=$PWD/audio-resampler_amd/lib/libdrb_audio_converter_amd_stamps.so timeout -k 10 120 python3 tools/stamp_split.py > $O/stamps_headline.txt 2>&1 || exit 1
