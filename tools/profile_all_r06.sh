#!/bin/bash
# Round-6 profile capture, ON THE GPU BOX from the repo root: the default bench line, the CR-CED kernel in its product form (v3_l2x6 = 3)
# and the comparator (0 = every layer on the fp32 MFMA, through the option's environment default), R-CED V1 / V2 fp32, config 2 (bf16,
# one launch: kernels_frame16.h), the training step, the audio kernels.  Outputs under gpurun_out/prof_r06* (the summaries are copied into
# profiles/ by hand, ONE capture commit).
set -u
python3 __graft_entry__.py > /dev/null 2>&1
python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python3 bench.py --variant 2 --dtype bf16 --batch 64 --steps 200 --cpu-seconds 0 --no-secondary > gpurun_out/r06_bench_config2.json 2> /dev/null
bash tools/profile.sh r06v3 > /dev/null 2>&1
TRACE_STEPS=300 TRACE_WARMUP=100 bash tools/profile.sh r06c2 --variant 2 --dtype bf16 --batch 64 > /dev/null 2>&1   # a 0.5-ms kernel: 13 launches run on idle clocks
bash tools/profile_train.sh r06 > /dev/null 2>&1
bash tools/profile_audio.sh r06 > /dev/null 2>&1
RCED_V3_L2X6=0 bash tools/profile.sh r06v3f32 > /dev/null 2>&1
bash tools/profile.sh r06v1 --variant 1 > /dev/null 2>&1
bash tools/profile.sh r06v2 --variant 2 > /dev/null 2>&1
for t in r06v3 r06c2 r06v3f32 r06v1 r06v2; do echo "== $t"; head -14 gpurun_out/prof_$t/summary.txt; done
head -30 gpurun_out/proft_r06/summary.txt
