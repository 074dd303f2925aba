#!/bin/bash
# generic tile kernels: XCD-aware tile order against the plain round-robin (variant library built with -DCFFT_TILE_XCD_ORDER=0)
S=c20736,c46080,c245760,c655360,c1000000,c1327104,c2097152,c34992
for lib in "" $PWD/chowdsp_fft_amd/variants/lib_noxcd.so; do
  echo "== ${lib:-library default (XCD order)}"
  CHOWDSP_FFT_LIB=$lib python tools/all_sizes_rate.py $S | cut -c1-100
  CHOWDSP_FFT_LIB=$lib python tools/rate_2d.py 256x256,1024x1024,2048x2048,720x1280,1080x1920 | cut -c1-100
done
