#!/bin/bash
# Run on the GPU box: rocprofv3 passes behind profiles/r02_*.  usage: bash scratch/profile_round.sh <tag>
# Every pass profiles `python3 bench.py ...` directly (no env / sh -c hop under rocprofv3); counters are collected in
# their own runs (--pmc with --kernel-trace only).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
COMMON="--no-cpu-baseline --no-other-configs --sustained-steps 0"
# round 6: the engine runs the weight gradients on a second stream beside the data-gradient chain; a kernel listed while
# another one shares the chip shows the pair's time.  One trace of the schedule AS IT RUNS first (wall against kernel sum),
# then every per-kernel pass with DSPN_WGRAD_SIDE=0: the kernels alone, as bench.py's instrumented steps time them
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_beside -o kt -- python3 bench.py --steps 6 --warmup 2 $COMMON > $OUT/bench_beside.log 2>&1
grep '^{"metric"' $OUT/bench_beside.log > profiles/${TAG}_f16x2_beside_bench_line.json
T=$(ls $OUT/kt_beside/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 scratch/step_profile_csv.py "$T" 40 > profiles/${TAG}_f16x2_beside_last_step_by_kernel.txt
export DSPN_WGRAD_SIDE=0
# f16x2 = the default math (fp32 results from two fp16 pieces, DSPN_MATH_F32_F16X2), x3 = three bf16 pieces (round 2's default),
# fp32 = fp32 MFMA, bf16 = bf16 tensors in HBM
for MODE in f16x2 x3 fp32 bf16; do
  ST=""; [ $MODE = bf16 ] && ST="--store bf16"; [ $MODE = fp32 ] && ST="--math fp32"; [ $MODE = x3 ] && ST="--math bf16x3"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$MODE -o kt -- python3 bench.py --steps 6 --warmup 2 $COMMON $ST > $OUT/bench_$MODE.log 2>&1
  grep '^{"metric"' $OUT/bench_$MODE.log > profiles/${TAG}_${MODE}_bench_line.json
  S=$(ls $OUT/kt_$MODE/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] && head -60 "$S" > profiles/${TAG}_${MODE}_kernel_stats_bench.csv
  T=$(ls $OUT/kt_$MODE/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python3 scratch/step_profile_csv.py "$T" 70 > profiles/${TAG}_${MODE}_last_step_by_kernel.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$MODE -o p -- python3 bench.py --steps 2 --warmup 1 --no-roofline $COMMON $ST > $OUT/pmc_fetch_$MODE.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$MODE -o p -- python3 bench.py --steps 2 --warmup 1 --no-roofline $COMMON $ST > $OUT/pmc_write_$MODE.log 2>&1
  python3 scratch/pmc_traffic2.py $OUT/pmc_fetch_$MODE $OUT/pmc_write_$MODE 3 profiles/${TAG}_${MODE}_pmc_hbm_traffic.csv profiles/${TAG}_${MODE}_pmc_conv_family.json > $OUT/pmc_$MODE.txt 2>&1
  cat $OUT/pmc_$MODE.txt
done
for MODE in f16x2 x3 fp32; do
  ST=""; [ $MODE = fp32 ] && ST="--math fp32"; [ $MODE = x3 ] && ST="--math bf16x3"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_$MODE -o p -- python3 bench.py --steps 2 --warmup 1 --no-roofline $COMMON $ST > $OUT/pmc_mfma_$MODE.log 2>&1
  python3 scratch/pmc_mfma2.py $OUT/pmc_mfma_$MODE > profiles/${TAG}_${MODE}_pmc_mfma_busy.csv 2>$OUT/pmc_mfma_$MODE.err
done
# keep the merged scratch small
find $OUT -name "*.csv" -size +8M -delete
# profiles/ does not travel back from the GPU box, gpurun_out/ does
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
ls -la profiles | grep $TAG
