#!/bin/bash
# static tile kernels: workgroup size for the lengths whose images leave room for two or three workgroups per CU
S=c245760,c61440,c82944,c57600,c23040,c38400
echo "== mid 1024 (library default)"; python tools/all_sizes_rate.py $S | cut -c1-100
for v in mid512 mid256; do echo "== $v"; CHOWDSP_FFT_LIB=$PWD/chowdsp_fft_amd/variants/lib_$v.so python tools/all_sizes_rate.py $S | cut -c1-100; done
