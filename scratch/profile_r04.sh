#!/bin/bash
# round-4 profile set (GPU box): the per-mode passes of profile_round.sh, the layer table, the coexec counters and the
# in-kernel phase stamps of the final build
bash scratch/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1
python scratch/layer_bench.py 32 > gpurun_out/profiles_r04/r04_f16x2_layer_table.txt 2>&1
bash scratch/pmc_coexec.sh r04_final > gpurun_out/coexec_r04_final.log 2>&1
cp gpurun_out/coexec_r04_final/a.csv gpurun_out/profiles_r04/r04_f16x2_coexec_final.csv
cp gpurun_out/coexec_r04_final/b.csv gpurun_out/profiles_r04/r04_f16x2_coexec_lds_insts_final.csv
python bench.py --mode infer --no-cpu-baseline > gpurun_out/profiles_r04/r04_infer_bench_line.json 2>/dev/null
ls gpurun_out/profiles_r04
