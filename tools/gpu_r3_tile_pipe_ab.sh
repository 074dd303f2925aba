S=c20736,c46080,c245760,r41472,c30000,c2097152
for v in 0 1 2 3; do echo "== TILE_PIPE=$v"; CHOWDSP_FFT_TILE_PIPE=$v python tools/all_sizes_rate.py $S | cut -c1-100; done
echo "== default"; python tools/all_sizes_rate.py $S | cut -c1-100
