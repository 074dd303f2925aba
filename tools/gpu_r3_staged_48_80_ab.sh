S=c48,c80,r96,r160,c96
for rep in 1 2; do
for v in "" st4880; do
  echo "== ${v:-default} rep $rep"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done; done
