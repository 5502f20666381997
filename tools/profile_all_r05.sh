#!/bin/bash
# Round-5 profile capture, ON THE GPU BOX from the repo root: the CR-CED kernel in its product form (v3_l2x6 = 3) and the comparators
# (2 = round 4's product, 0 = every layer on the fp32 MFMA, through the option's environment default), R-CED V1 / V2 fp32, config 2 bf16,
# the training step, the audio kernels.  Outputs under gpurun_out/prof_r05* (summaries are copied into profiles/ by hand, ONE capture commit).
set -u
bash tools/profile.sh r05v3 > /dev/null 2>&1
RCED_V3_L2X6=2 bash tools/profile.sh r05v3form2 > /dev/null 2>&1
RCED_V3_L2X6=0 bash tools/profile.sh r05v3f32 > /dev/null 2>&1
bash tools/profile.sh r05v1 --variant 1 > /dev/null 2>&1
bash tools/profile.sh r05v2 --variant 2 > /dev/null 2>&1
bash tools/profile.sh r05c2 --variant 2 --dtype bf16 --batch 64 > /dev/null 2>&1
bash tools/profile_train.sh r05 > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r05audio" -- python3 "$GRAFT_REPO_ROOT/tools/bench_audio.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_r05audio.log" 2>&1 )
for t in r05v3 r05v3form2 r05v3f32 r05v1 r05v2 r05c2; do echo "== $t"; head -14 gpurun_out/prof_$t/summary.txt; done
head -30 gpurun_out/proft_r05/summary.txt
