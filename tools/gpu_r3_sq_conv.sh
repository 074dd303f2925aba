#!/bin/bash
# round 3: SQ counters of the fused-convolution and engine kernels (none existed), next to the plain transforms of the same size
REPO=$PWD
mkdir -p gpurun_out/r3
{
echo "##### conv_real_n4096_b131072 (conv_frames_real_kernel_pf)"
tools/prof_sq_cmd.sh r3/sq_conv_real4096 python3 $REPO/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-secondary --workload conv_real_n4096_b131072
echo "##### plain real N=4096 r2c + c2r (tools/all_sizes_rate.py r4096)"
tools/prof_sq_cmd.sh r3/sq_real4096 python3 $REPO/tools/all_sizes_rate.py r4096
echo "##### ols_engine_b2048_p16_c8192 (ols_block_kernel)"
tools/prof_sq_cmd.sh r3/sq_ols python3 $REPO/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-secondary --workload ols_engine_b2048_p16_c8192
echo "##### split-plane fused convolution c8192 c16384 c32768 r65536 (conv_frames_*_split_kernel)"
tools/prof_sq_cmd.sh r3/sq_conv_split python3 $REPO/tools/conv_frames_rate.py c8192 c16384 c32768 r65536
echo "##### mixed-radix fused convolution c480 c9216 (generic_conv_kernel / generic_inplace_conv_kernel)"
tools/prof_sq_cmd.sh r3/sq_conv_generic python3 $REPO/tools/conv_frames_rate.py c480 c9216
} > gpurun_out/r3/sq_counters_conv.txt 2>&1
tail -40 gpurun_out/r3/sq_counters_conv.txt
