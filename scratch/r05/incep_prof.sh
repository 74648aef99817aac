#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--network inceptionv3 --size 512 --width 1024 --batch 8 --store bf16 --no-cpu-baseline --no-other-configs"
python3 bench.py $ARGS --steps 20 --warmup 5 | tail -1 > gpurun_out/incep_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/incep_prof -o kt -- python3 bench.py $ARGS --steps 6 --warmup 2 > gpurun_out/incep_prof.log 2>&1
T=$(ls gpurun_out/incep_prof/*kernel_trace.csv | head -1)
python3 scratch/step_profile_csv.py "$T" 60 > gpurun_out/incep_last_step.txt
head -70 gpurun_out/incep_last_step.txt
python3 -c "
import json; d=json.load(open('gpurun_out/incep_line.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
find gpurun_out/incep_prof -name '*.csv' -size +8M -delete
