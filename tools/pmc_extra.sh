#!/bin/bash
# Run ON THE GPU BOX: extra PMC passes of the default bench workload (instruction cache, LDS queues).
# Usage: tools/pmc_extra.sh <tag> [extra bench args]   -> gpurun_out/pmcx_<tag>/
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --no-profile --no-secondary $*"
pmc() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_$name.log" 2>&1
}
pmc icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL
pmc ldsq SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_VALU_MFMA_COEXEC_CYCLES
pmc act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused" in r.get("Kernel_Name", "") or "gemm" in r.get("Kernel_Name", ""):
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"] or 0); acc[k][1] += 1
for (k, c), (t, n) in sorted(acc.items()):
    print("%-42s %-30s avg=%.6g (n=%d)" % (k, c, t / n, n))
PY
