#!/bin/bash
# per-kernel time of the last step with the tile-spanning loop off / on (rocprofv3 kernel trace of bench.py)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for X in 0 1; do
  export DSPN_XT=$X
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/kt_xt$X -o kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --sustained-steps 0 > gpurun_out/r06/bench_xt$X.log 2>&1
  T=$(ls gpurun_out/r06/kt_xt$X/*kernel_trace.csv 2>/dev/null | head -1)
  python3 scratch/step_profile_csv.py "$T" 70 > gpurun_out/r06/last_step_xt$X.txt
  find gpurun_out/r06/kt_xt$X -name "*.csv" -size +8M -delete
done
head -30 gpurun_out/r06/last_step_xt1.txt
