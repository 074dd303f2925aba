#!/bin/bash
# Variant of the library that differs from the current build in ONE translation unit (seconds instead of minutes):
# usage: tools/build_tu_variant.sh <name> <tu, e.g. kernels_group_real> <packed 0|1> "<extra flags>"
# Output: chowdsp_fft_amd/variants/lib_<name>.so (tools/time_variants.py); the other objects are the main build's (run make first).
set -e
NAME=$1; TU=$2; PACKED=$3; FLAGS=$4
cd "$(dirname "$0")/../chowdsp_fft_amd/csrc"
mkdir -p ../variants
OBJS=$(ls *.o | grep -v "^$TU.o$")
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function -fvisibility=hidden -ffp-contract=fast -DCFFT_PACKED=$PACKED $FLAGS -c $TU.hip -o ../variants/${TU}_${NAME}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_${NAME}.so $OBJS ../variants/${TU}_${NAME}.o
ls -la ../variants/lib_${NAME}.so
