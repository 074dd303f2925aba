#!/bin/bash
# two-stage plans that cannot store whole lines with their sorted radices: 48 = 16 x 3 and 80 = 16 x 5 instead of 6 x 8 / 10 x 8
# (variant -DCFFT_TWO_STAGE_16_FIRST=1), and 96 = 16 x 6 / 144 = 16 x 9 with direct instead of staged I/O (= 2)
S=c48,c80,c96,c144,r96,r160,r192,r288
for rep in 1 2; do
for v in "" f16a f16b; do
  echo "== ${v:-default} rep $rep"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done; done
