#!/bin/bash
# Round-3 profile capture, ON THE GPU BOX from the repo root: forward kernels (CR-CED, R-CED V1 / V2 fp32, config 2 bf16),
# the training step, and the FETCH_SIZE calibration.  Outputs under gpurun_out/ (copied into profiles/ by hand).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile.sh r03v3 > /dev/null 2>&1
bash tools/profile.sh r03v1 --variant 1 > /dev/null 2>&1
bash tools/profile.sh r03v2 --variant 2 > /dev/null 2>&1
bash tools/profile.sh r03c2 --variant 2 --dtype bf16 --batch 64 > /dev/null 2>&1
bash tools/profile_train.sh r03 > /dev/null 2>&1
mkdir -p gpurun_out/fetch_cal && cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_cal $ROOT/tools/micro/fetch_cal.hip > $ROOT/gpurun_out/fetch_cal/build.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/fetch_cal/pmc -- /tmp/fetch_cal > $ROOT/gpurun_out/fetch_cal/run.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("gpurun_out/fetch_cal/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE" and "read_stream" in r.get("Kernel_Name", ""):
            k = r["Kernel_Name"].split("(")[0]
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-40s FETCH_SIZE KiB per launch: %s  -> bytes/known = %.3f" % (k, ["%.0f" % x for x in v], sum(v) / len(v) * 1024 / 67633152))
PY
