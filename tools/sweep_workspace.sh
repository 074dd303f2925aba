#!/bin/bash
# two-pass (N=65536) throughput vs workspace cap (total over lanes) and lane count
for LANES in 1 2; do for MIB in 128 256 512; do
  echo "== lanes=$LANES CHOWDSP_FFT_WORKSPACE_MIB=$MIB"
  CHOWDSP_FFT_LANES=$LANES CHOWDSP_FFT_WORKSPACE_MIB=$MIB python tools/time_variants.py --workloads c65536 --rounds 3 chowdsp_fft_amd/libchowdsp_fft_hip.so 2>&1 | grep -E "median"
done; done
