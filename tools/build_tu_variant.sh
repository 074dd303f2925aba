#!/bin/bash
# Variant of the library that differs from the current build in ONE translation unit (seconds instead of minutes):
# usage: tools/build_tu_variant.sh <name> <tu, e.g. kernels_group_real> "<extra flags>"
# The object is compiled THROUGH the Makefile (same CXXFLAGS, same per-unit CFFT_PACKED value: one place to change either), into
# chowdsp_fft_amd/variants/obj_<name>/, and linked with the main build's other objects (run make first).
# Output: chowdsp_fft_amd/variants/lib_<name>.so (tools/time_variants.py).
set -e
NAME=$1; TU=$2; FLAGS=$3
if [ "$FLAGS" = "0" ] || [ "$FLAGS" = "1" ]; then FLAGS=$4; fi   # (old call form with the packed value as third argument: now taken from the Makefile)
cd "$(dirname "$0")/../chowdsp_fft_amd/csrc"
mkdir -p ../variants
make OBJDIR="../variants/obj_${NAME}" EXTRA="$FLAGS" "../variants/obj_${NAME}/${TU}.o"
OBJS=$(ls *.o | grep -v "^$TU.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/lib_${NAME}.so $OBJS ../variants/obj_${NAME}/${TU}.o
ls -la ../variants/lib_${NAME}.so
