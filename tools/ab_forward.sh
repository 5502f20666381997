#!/bin/bash
# A/B over forward configs: product vs exp/sd.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in $R/fullycnnspeechenhancement_amd/librced_hip.so $R/exp/sd.so; do
  for cfg in "--variant 3" "--variant 1" "--variant 2" "--variant 2 --dtype bf16 --batch 64" "--variant 1 --dtype bf16 --batch 64"; do
    RCED_LIB=$L python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-20s %-40s ms/step=%.3f fused_ms=%.3f Mfps=%.2f' % ('$(basename $L)', '$cfg', d['ms_per_step'], r['avg_launch_ms'], d['value']/1e6))"
  done
done
