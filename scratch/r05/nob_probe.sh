#!/bin/bash
# timing experiment: conv_ntv_kernel with every weight request out of range (libdspn_hip_var.so built with -DDSPN_NTV_ABL_NOB)
python scratch/layer_bench.py 32 2>&1 | grep -E "conv1|_sc |TOTAL" > gpurun_out/nob_real.txt
DSPN_LIB=$PWD/dspnet_amd/libdspn_hip_var.so python scratch/layer_bench.py 32 2>&1 | grep -E "conv1|_sc |TOTAL" > gpurun_out/nob_abl.txt
paste -d'\n' gpurun_out/nob_real.txt gpurun_out/nob_abl.txt | cut -c1-120
