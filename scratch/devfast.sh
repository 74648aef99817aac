#!/bin/bash
# compile conv.hip with only the 8-wave two-piece kernels (-DDSPN_DEV_FAST) and print their registers / scratch
cd /root/repo/dspnet_amd/csrc && time /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function -DDSPN_DEV_FAST $1 -c conv.hip -o /tmp/t/conv_fast.o 2>&1 | grep -E "error|real" -A3
cd /root/repo; python3 scratch/kres_obj.py /tmp/t/conv_fast.o conv_nt_kernel | grep "^4, 2, 1, 2, true, 3"
