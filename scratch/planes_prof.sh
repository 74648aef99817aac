cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 1; do
export DSPN_DY_PLANES=$m
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_pl$m -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline > gpurun_out/kt_pl$m.log 2>&1
python3 scratch/step_profile_csv.py $(ls gpurun_out/kt_pl$m/*kernel_trace.csv | head -1) 40 > gpurun_out/last_step_pl$m.txt
find gpurun_out/kt_pl$m -name "*.csv" -size +8M -delete
done
paste <(head -32 gpurun_out/last_step_pl0.txt | cut -c1-100) <(head -32 gpurun_out/last_step_pl1.txt | cut -c1-100)
