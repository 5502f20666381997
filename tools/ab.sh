#!/bin/bash
# A/B timing of alternative builds of the same ABI (exp/*.so), one bench process each.
# Usage (on the GPU box): tools/ab.sh lib1.so lib2.so ...   (default: product lib + all exp/*.so)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS="$@"; [ -z "$LIBS" ] && LIBS="$ROOT/fullycnnspeechenhancement_amd/librced_hip.so $(ls $ROOT/exp/*.so 2>/dev/null)"
for L in $LIBS; do
  RCED_LIB=$L python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-50s ms/step=%.3f fused_ms=%.3f other=%s' % ('$(basename $L)', d['ms_per_step'], r['avg_launch_ms'], r['other_kernels_ms_per_step']))"
done
