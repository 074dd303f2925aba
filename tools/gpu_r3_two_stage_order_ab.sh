#!/bin/bash
# two-stage mixed-radix plans (160, 192, 240 points): last stage storing full 128-byte lines (library default: 16 x 10 / 16 x 12 /
# 16 x 15) against the sorted order (variant -DCFFT_TWO_STAGE_FULL_LINE=0); and n = 160 with direct instead of staged I/O
S=c160,c192,c240,r320,r384,r480,c96,c144,c20736
for rep in 1 2; do
for v in "" sorted2 direct160; do
  echo "== ${v:-default} rep $rep"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done; done
