#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel-trace stats + PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [extra bench args]     -> gpurun_out/prof_<tag>/...
# Counters go in separate passes with kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/__graft_entry__.py > /dev/null 2>&1   # build first, in a process of its own (never under the profiler)
BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --no-profile --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps ${TRACE_STEPS:-10} --warmup ${TRACE_WARMUP:-3} > "$OUT/trace.log" 2>&1
pmc() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_$name.log" 2>&1
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA
cd "$ROOT"
python3 tools/summarize_prof.py "$OUT" "$OUT/pmc_traffic.json" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
