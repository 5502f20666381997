#!/bin/bash
# Run ON THE GPU BOX from the repo root: rocprofv3 kernel stats + PMC passes of the training step (tools/bench_train.py).
# Usage: tools/profile_train.sh <tag>   -> gpurun_out/proft_<tag>/summary.txt   (counters in separate passes, kernel-trace only)
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/proft_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/__graft_entry__.py > /dev/null 2>&1   # build first, never under the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/bench_train.py > "$OUT/trace.log" 2>&1
pmc() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/tools/bench_train.py > "$OUT/pmc_$name.log" 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA
cd "$ROOT"
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def find(d, pat): return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
def short(n): return n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('rced::', '')[:62]
print("# rocprofv3, tools/bench_train.py (CR-CED V3 train step, batch 256 x 512, 4 steps)")
for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print("## kernel-trace --stats, per step (4 steps)")
    for r in rows[:26]:
        print('%-62s calls/step %5.1f  ms/step %7.2f  avg %7.3f' % (short(r['Name']), int(r['Calls']) / 4, float(r['TotalDurationNs']) / 4e6, float(r['AverageNs']) / 1e6))
    print('total ms/step %.2f' % (tot / 4e6))
print("## PMC, average per dispatch")
acc = defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = (short(r.get("Kernel_Name", "")), r.get("Counter_Name"))
            acc[k][0] += float(r.get("Counter_Value", 0) or 0); acc[k][1] += 1
# HBM bytes per step: FETCH_SIZE + WRITE_SIZE (KiB) summed over every dispatch of the 4 profiled steps / 4.  FETCH_SIZE is
# doubled (MI355X_MICROARCH.md, HBM: wide coalesced streaming reads are counted at half on gfx950; the training kernels
# stage their tiles with 16-byte loads / LDS-DMA).
import json
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for (k, c), (v, n) in acc.items():
    if c in tot: tot[c] += v
gb = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 4 / 1e9
json.dump({"_comment": "CR-CED V3 train step, batch 256 x 512: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) summed over "
                       "all kernels of a step; FETCH_SIZE doubled: on gfx950 the counter reports half the bytes of a streaming read for every access width (profiles/r03_fetch_calibration.txt)",
           "kernel_hash": ge.train_kernel_hash(), "fetch_kib_raw_per_step": tot["FETCH_SIZE"] / 4, "write_kib_per_step": tot["WRITE_SIZE"] / 4,
           "hbm_gb_per_step": gb}, open(os.path.join(out, "pmc_train.json"), "w"), indent=1)
print("HBM per step: %.1f GB (fetch x2 + write)" % gb)
kern = sorted({k for k, _ in acc})
for k in kern:
    if not any(s in k for s in ("mfma", "bwd_route2", "first_", "final_", "bn_act")): continue
    print(k)
    print("   " + "  ".join("%s=%.4g" % (c, acc[(kk, c)][0] / max(acc[(kk, c)][1], 1)) for (kk, c) in sorted(acc) if kk == k))
PY
cat "$OUT/summary.txt"
