#!/bin/bash
# whole-frame mixed-radix plans: a radix moved to the end so that the last stage stores whole 128-byte lines (library) against the
# sorted lists (chowdsp_fft_amd/variants/lib_oldorder.so = the library one commit earlier)
S=c288,c320,c400,c432,c480,c576,c720,c800,c864,c960,c1600,c1920,c2000,r576,r640,r960,r1920,r3840,c2400,c3840,c384,c768
for rep in 1 2; do
for v in "" oldorder; do
  echo "== ${v:-default} rep $rep"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done; done
