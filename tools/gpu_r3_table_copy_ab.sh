S=c96,c384,c768,c1536,c1920,c5120,c19200,r96,r384,r1920,r3840,c480,c640
for rep in 1 2; do
for v in "" copy1; do
  echo "== ${v:-default} rep $rep"
  CHOWDSP_FFT_LIB=${v:+$PWD/chowdsp_fft_amd/variants/lib_$v.so} python tools/all_sizes_rate.py $S | cut -c1-100
done; done
