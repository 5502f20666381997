#!/bin/bash
# Round-4 profile capture, ON THE GPU BOX from the repo root: the CR-CED kernel in its three forms (option v3_l2x6 = 2 / 1 / 0,
# the latter two through the option's environment default), R-CED V1 / V2 fp32, config 2 bf16, and the training step.
# Outputs under gpurun_out/prof_r04* (summaries are copied into profiles/ by hand).
set -u
bash tools/profile.sh r04v3 > /dev/null 2>&1
RCED_V3_L2X6=1 bash tools/profile.sh r04v3x6 > /dev/null 2>&1
RCED_V3_L2X6=0 bash tools/profile.sh r04v3f32 > /dev/null 2>&1
bash tools/profile.sh r04v1 --variant 1 > /dev/null 2>&1
bash tools/profile.sh r04v2 --variant 2 > /dev/null 2>&1
bash tools/profile.sh r04c2 --variant 2 --dtype bf16 --batch 64 > /dev/null 2>&1
bash tools/profile_train.sh r04 > /dev/null 2>&1
for t in r04v3 r04v3x6 r04v3f32 r04v1 r04v2 r04c2; do echo "== $t"; head -12 gpurun_out/prof_$t/summary.txt; done
head -30 gpurun_out/proft_r04/summary.txt
