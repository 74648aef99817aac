#!/bin/bash
# the two HBM-traffic counter passes of profile_round.sh alone (after pmc_traffic2.py learnt the wide family's kernel names)
set -u
TAG=r05
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
COMMON="--no-cpu-baseline --no-other-configs"
for MODE in f16x2 x3 fp32 bf16; do
  ST=""; [ $MODE = bf16 ] && ST="--store bf16"; [ $MODE = fp32 ] && ST="--math fp32"; [ $MODE = x3 ] && ST="--math bf16x3"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODE -o p -- python3 bench.py --steps 2 --warmup 1 --no-roofline $COMMON $ST > $OUT/pmc_fetch_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODE -o p -- python3 bench.py --steps 2 --warmup 1 --no-roofline $COMMON $ST > $OUT/pmc_write_$MODE.log 2>&1
  python3 scratch/pmc_traffic2.py $OUT/pmc_fetch_$MODE $OUT/pmc_write_$MODE 3 profiles/${TAG}_${MODE}_pmc_hbm_traffic.csv profiles/${TAG}_${MODE}_pmc_conv_family.json > $OUT/pmc_$MODE.txt 2>&1
  cat $OUT/pmc_$MODE.txt
done
find $OUT -name "*.csv" -size +8M -delete
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_*pmc_hbm* profiles/${TAG}_*pmc_conv* gpurun_out/profiles_$TAG/
