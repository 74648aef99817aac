#!/bin/bash
# conv_ntv_kernel<4,2> (256 x 128 on eight waves, one workgroup per CU) against the automatic choice, per layer
python scratch/layer_bench.py 32 2>&1 | grep -E "conv1|_sc |conv3|TOTAL" > gpurun_out/ntv42_auto.txt
DSPN_WIDE_TILES=2 python scratch/layer_bench.py 32 2>&1 | grep -E "conv1|_sc |conv3|TOTAL" > gpurun_out/ntv42_forced.txt
paste -d'\n' gpurun_out/ntv42_auto.txt gpurun_out/ntv42_forced.txt | cut -c1-125
