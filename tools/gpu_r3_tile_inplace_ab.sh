#!/bin/bash
# static tile kernels: one LDS image with in-place stages from n = 324 (library default) against from n = 640 only (variant
# library -DCFFT_TILE_INPLACE_MIN_N=640: two images and 1024 work-items for n = 324 .. 639 as before)
S=c20736,c46080,c245760,c61440,c655360,c82944,c230400,c192000,c331776,c129600,c1327104,c1000000,c34992
for lib in "" $PWD/chowdsp_fft_amd/variants/lib_inpl640.so; do
  echo "== ${lib:-library default (in place from 324)}"
  CHOWDSP_FFT_LIB=$lib python tools/all_sizes_rate.py $S | cut -c1-100
  CHOWDSP_FFT_LIB=$lib python tools/rate_2d.py 256x256,512x512,480x640,600x800,1024x1024,720x1280,1080x1920 | cut -c1-100
done
