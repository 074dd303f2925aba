#!/bin/bash
# power-of-two tile kernels of the mixed two-pass plans (pass_a_kernel / pass_b_kernel): loads in flight per work-item in the tile
# loops, 4 (library) against 8 / 16 (variants -DCFFT_POW2_TILE_UNROLL=8 / 16)
S=c245760,c61440,c655360,c24576,c98304,c122880,c491520,c20480
for rep in 1 2; do
for v in "" un8 un16; do
  echo "== ${v:-default} rep $rep"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done; done
