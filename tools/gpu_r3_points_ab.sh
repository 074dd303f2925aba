#!/bin/bash
# points in flight per workgroup of the two-image mixed-radix kernel, after the radix lists were re-ordered: the per-length choice of
# generic_geometry (library) against 1536 / 2048 / 3072 everywhere (variants -DCFFT_GENERIC_POINTS=... -DCFFT_GENERIC_POINTS_TUNED=0)
S=c48,c80,c96,c144,c160,c192,c240,c288,c320,c384,c400,c432,c480,c576,c640,c720,c768,c800,c864,c960,r96,r192,r288,r384,r480,r576,r640,r768,r960,r1152,r1280,r1920
for v in "" p1536 p2048 p3072; do
  echo "== ${v:-default}"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done
echo "== default again"
python tools/all_sizes_rate.py $S | cut -c1-100
