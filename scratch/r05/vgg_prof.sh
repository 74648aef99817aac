#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--network vgg16_reduced --batch 16 --no-cpu-baseline --no-other-configs"
python3 bench.py $ARGS --steps 10 --warmup 3 | tail -1 > gpurun_out/vgg_line.json
python3 bench.py $ARGS --steps 10 --warmup 3 --wide-tiles 1 | tail -1 > gpurun_out/vgg_line_narrow.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vgg_prof -o kt -- python3 bench.py $ARGS --steps 4 --warmup 2 > gpurun_out/vgg_prof.log 2>&1
T=$(ls gpurun_out/vgg_prof/*kernel_trace.csv | head -1)
python3 scratch/step_profile_csv.py "$T" 40 > gpurun_out/vgg_last_step.txt
head -45 gpurun_out/vgg_last_step.txt
python3 -c "
import json
for f in ('vgg_line','vgg_line_narrow'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])"
find gpurun_out/vgg_prof -name '*.csv' -size +8M -delete
