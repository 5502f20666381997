#!/bin/bash
# Per-kernel A/B of the training step over alternative builds (exp/*.so): rocprofv3 kernel stats per lib.
# Usage (on the GPU box, repo root): tools/ab_train.sh name1 name2 ...   -> gpurun_out/abt_<name>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  export RCED_LIB=$R/exp/$n.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abt_$n -- python3 $R/tools/bench_train.py > $R/gpurun_out/abt_$n.log 2>&1
  f=$(find $R/gpurun_out/abt_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n: $(grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/abt_$n.log)"
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mfma' in r['Name']:
        print('  %-60s %8.3f ms' % (r['Name'].split('(')[0][-60:], float(r['AverageNs'])/1e6))
PY
done
