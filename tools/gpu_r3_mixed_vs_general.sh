#!/bin/bash
# long mixed-radix lengths with a power-of-two row factor: generic columns x power-of-two rows (library default) against the general
# form on two static tile lengths (CHOWDSP_FFT_MIXED_TO_GENERAL=1)
S=c245760,c61440,c655360,c24576,c98304,c30720,c20480,c76800,c1310720,c122880,c491520
for v in 0 1; do echo "== MIXED_TO_GENERAL=$v"; CHOWDSP_FFT_MIXED_TO_GENERAL=$v python tools/all_sizes_rate.py $S | cut -c1-100; done
