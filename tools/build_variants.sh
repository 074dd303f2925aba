#!/bin/bash
# Build differently-flagged copies of the library for A/B timing on the GPU box (tools/time_variants.py).
# usage: tools/build_variants.sh name1="-DFLAG=1 ..." name2="..."      (run in the build container)
# Output: chowdsp_fft_amd/variants/lib_<name>.so   (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/../chowdsp_fft_amd/csrc"
mkdir -p ../variants
for spec in "$@"; do
  name="${spec%%=*}"
  flags="${spec#*=}"
  echo "== variant $name : $flags"
  make -j8 LIBNAME="variants/lib_${name}.so" OBJDIR="../variants/obj_${name}" EXTRA="$flags" 2>&1 | grep -E "error|warning: v" || true
done
ls -la ../variants/*.so
