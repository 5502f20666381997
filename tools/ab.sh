#!/bin/bash
# A/B of alternative builds of the same ABI on the default bench workload: tools/ab.sh [name ...]  (exp/<name>.so; "prod" = the product)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  if [ "$n" = prod ]; then L=$R/fullycnnspeechenhancement_amd/librced_hip.so; else L=$R/exp/$n.so; fi
  RCED_LIB=$L python3 $R/bench.py --steps ${STEPS:-40} --warmup 10 --cpu-seconds 0 --no-secondary $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s ms/step=%.3f fused_ms=%.3f frac=%.4f Mfps=%.2f other=%s' % ('$n', d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['value']/1e6, r['other_kernels_ms_per_step']))"
done
