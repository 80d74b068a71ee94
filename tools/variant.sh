#!/bin/bash
# Builds the library with extra compiler flags into audio-resampler_amd/lib/<name>/
# (for tools/try.sh LIBS="<name> ..."): tools/variant.sh <name> [-DMACRO ...]
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
B=$ROOT/audio-resampler_amd/build/$NAME
mkdir -p $B $ROOT/audio-resampler_amd/lib/$NAME
S=$ROOT/audio-resampler_amd/csrc
for f in drb_plan drb_batch drb_api; do gcc -std=gnu11 -O2 -fPIC -I$ROOT/include -I$S "$@" -c -o $B/$f.o $S/$f.c; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -I$S -Wno-unused-function -c -o $B/drb_kernels.o $S/drb_kernels.hip &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I$S -Wno-unused-function "$@" -c -o $B/drb_kernels_fast.o $S/drb_kernels_fast.hip
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/audio-resampler_amd/lib/$NAME/libdrb_audio_converter_amd.so $B/*.o -lpthread
