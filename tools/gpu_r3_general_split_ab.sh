#!/bin/bash
# general two-pass plans: which pair of static lengths, and which of the two is the column length n2 (CHOWDSP_FFT_GENERAL_SPLIT=n1xn2)
run() { echo "== $1 split ${2:-default}"; CHOWDSP_FFT_GENERAL_SPLIT=$2 python tools/all_sizes_rate.py $1 | cut -c1-100; }
for rep in 1 2; do
run c48000 240x200; run c48000 200x240
run c100000 400x250; run c100000 250x400
run c76800 240x320; run c76800 320x240
run c192000 400x480; run c192000 480x400
run c34992 243x144; run c34992 216x162; run c34992 162x216
run c38400 240x160; run c38400 160x240; run c38400 200x192
run c62208 432x144; run c62208 243x256; run c62208 256x243
done
