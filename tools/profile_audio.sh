#!/bin/bash
# ON THE GPU BOX from the repo root: kernel-trace stats + PMC passes (separate, kernel-trace only) of the STFT / ISTFT kernels at
# config-3 scale (tools/bench_audio.py).  -> gpurun_out/prof_<tag>audio/summary.txt
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}audio
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/__graft_entry__.py > /dev/null 2>&1
B="python3 $ROOT/tools/bench_audio.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $B > "$OUT/trace.log" 2>&1
pmc() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $B > "$OUT/pmc_$name.log" 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA
pmc mem SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES
cd "$ROOT"
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
print("# rocprofv3, tools/bench_audio.py (STFT / ISTFT at config-3 scale: 256 utterances x 512 frames)")
for f in glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "stft" in r["Name"] or "deemph" in r["Name"]:
            print("%-70s calls=%s avg_ns=%s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "stft" in n:
            acc[n.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    v = {c: sum(x) / len(x) for c, x in d.items()}
    print("\n" + k)
    print("   " + "  ".join("%s=%.4g" % (c, v[c]) for c in sorted(v)))
    gui = v.get("GRBM_GUI_ACTIVE")
    if gui:
        cyc = gui / 8.0
        line = "   derived:"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v: line += " mfma_pipe_busy=%.1f%%" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc))
        if "SQ_LDS_IDX_ACTIVE" in v: line += " lds_active=%.1f%% (conflicts %.0f%% of it)" % (100 * v["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v["SQ_LDS_IDX_ACTIVE"], 1))
        if "SQ_WAVE_CYCLES" in v: line += " wait_any=%.0f%% wait_inst=%.0f%% active=%.0f%% of wave-cycles" % (100 * v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 100 * v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"])
        if "FETCH_SIZE" in v: line += " hbm_read=%.0f MB (x2 corrected) write=%.0f MB" % (2 * v["FETCH_SIZE"] / 1024, v.get("WRITE_SIZE", 0) / 1024)
        print(line)
PY
cat "$OUT/summary.txt"
