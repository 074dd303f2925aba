#!/bin/bash
# real pair stage of the generic kernels: pairs per work-item in flight (variant libraries -DCFFT_REAL_PAIR_BATCH=1 / 2 / 8; 4 = default)
S=r96,r192,r288,r384,r480,r640,r768,r1440,r1920,r2880,r3840,r4608,r9216,r12288,r18432
for v in "" pair1 pair2 pair8; do
  echo "== ${v:-library default (4)}"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done
