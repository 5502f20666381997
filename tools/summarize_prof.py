#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small text summary (what gets committed
under profiles/): per-kernel time stats from --kernel-trace --stats and per-kernel PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    for key in ("fused_v3_kernel", "final_gemm_kernel", "conv_layer_generic", "fused_v1", "fused_v2"):
        if key in name:
            return key
    return name[:60]


def traffic_json(out, path):
    """profiles/rNN_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE per launch (KiB) of the forward kernels, tagged with the
    hash of the kernel sources they were measured on (bench.py attaches roofline.traffic only on a match).
    FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: the counter reports half the bytes of streaming reads on
    gfx950).  The guide states that for 16 B/lane loads and calls 4 B/lane "uncalibrated"; tools/micro/fetch_cal.hip read a
    known 67,633,152 B with 4-, 8- and 16-byte lane loads and FETCH_SIZE came back as 33,034 KiB = 0.500 x the bytes for
    ALL three widths (round 3, profiles/r03_fetch_calibration.txt) -- so the fused kernel's 4 B/lane input stream gets
    the same factor (round 2 left it uncorrected and under-reported the kernel's reads by half)."""
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    vals = defaultdict(lambda: [0.0, 0])
    for d in glob.glob(os.path.join(out, "pmc_*")):
        for f in find(d, "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") in ("FETCH_SIZE", "WRITE_SIZE"):
                        k = (short(r.get("Kernel_Name", "")), r.get("Counter_Name"))
                        vals[k][0] += float(r.get("Counter_Value", 0) or 0)
                        vals[k][1] += 1
    rec = {"_comment": "HBM traffic per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB); "
                       "tools/profile.sh -> bench.py --steps 2 --warmup 1 at the default workload",
           "kernel_hash": ge.forward_kernel_hash(), "workload": {"variant": 3, "batch": 256, "frames": 512}}
    for kern in ("fused_v3_kernel", "final_gemm_kernel"):
        f, w = vals.get((kern, "FETCH_SIZE")), vals.get((kern, "WRITE_SIZE"))
        if f and w and f[1] and w[1]:
            raw = f[0] / f[1]
            rec[kern] = {"fetch_kib": 2 * raw, "fetch_kib_raw": raw, "write_kib": w[0] / w[1]}
    rec["fetch_calibration"] = {"factor": 2.0, "measured_counter_over_bytes": {"4B_per_lane": 0.5, "8B_per_lane": 0.5, "16B_per_lane": 0.5},
                                "source": "tools/micro/fetch_cal.hip under rocprofv3 --pmc FETCH_SIZE, profiles/r03_fetch_calibration.txt"}
    with open(path, "w") as fh:
        json.dump(rec, fh, indent=1)


def main(out):
    print("# rocprofv3 summary:", os.path.basename(out))
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        print("\n## kernel-trace --stats (%s)" % os.path.basename(f))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                n = row.get("Name", "")
                if any(k in n for k in ("rced", "fused", "final_gemm", "conv_layer")):
                    print("%-22s calls=%s total_ns=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
                        short(n), row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"),
                        row.get("MinNs"), row.get("MaxNs"), row.get("Percentage")))
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        seen = {}
        for r in rows:
            k = short(r.get("Kernel_Name", ""))
            if k not in seen and any(s in k for s in ("fused", "final", "conv_layer")):
                seen[k] = r
        print("\n## dispatch resources (kernel_trace.csv)")
        for k, r in seen.items():
            print("%-22s grid=%s wg=%s VGPR=%s accum=%s SGPR=%s LDS=%s scratch=%s" % (
                k, r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
                r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
    print("\n## PMC (average per dispatch)")
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for f in find(d, "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = (short(r.get("Kernel_Name", "")), r.get("Counter_Name"))
                    if any(s in k[0] for s in ("fused", "final", "conv_layer")):
                        acc[k][0] += float(r.get("Counter_Value", 0) or 0)
                        acc[k][1] += 1
        for (k, c), (tot, n) in sorted(acc.items()):
            print("%-22s %-28s avg=%.6g  (n=%d)" % (k, c, tot / max(n, 1), n))


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2])
