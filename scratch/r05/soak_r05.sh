#!/bin/bash
# round 5 stability evidence on one box: 300 SGD steps in the three fp32-result modes (losses finite and falling), and the
# side-stream race check at batch 32 -- two runs of the default schedule and one with the branch on the main stream must print
# the same parameter / gradient hashes line by line
python scratch/soak.py > gpurun_out/r05_soak_300_steps.txt 2>&1; tail -21 gpurun_out/r05_soak_300_steps.txt
python scratch/side_race_check.py 32 12 > gpurun_out/race_a.txt 2>&1
python scratch/side_race_check.py 32 12 > gpurun_out/race_b.txt 2>&1
DSPN_DET_SIDE=0 python scratch/side_race_check.py 32 12 > gpurun_out/race_main.txt 2>&1
(echo "# python scratch/side_race_check.py 32 12: default schedule twice, then DSPN_DET_SIDE=0 (branch on the main stream)"; grep -E "^[0-9]+ " gpurun_out/race_a.txt | tail -3; echo "identical a/b: $(cmp -s <(grep -E '^[0-9]+ ' gpurun_out/race_a.txt) <(grep -E '^[0-9]+ ' gpurun_out/race_b.txt) && echo yes || echo NO)"; echo "identical a/main-stream: $(cmp -s <(grep -E '^[0-9]+ ' gpurun_out/race_a.txt) <(grep -E '^[0-9]+ ' gpurun_out/race_main.txt) && echo yes || echo NO)") > gpurun_out/r05_side_race_check.txt
cat gpurun_out/r05_side_race_check.txt
