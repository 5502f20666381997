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
    for key in ("fused_v3_kernel", "final_gemm_kernel", "conv_layer_generic", "fused_v1", "fused_v2", "frame16_kernel"):
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


def code_object_resources():
    """{short kernel name: dict} from the amdhsa metadata of the shipped library: what the kernel-trace CSV does not carry (its VGPR_Count is
    the allocation granule count, its LDS_Block_Size the STATIC size: the fused kernels' LDS is dynamic)."""
    import re
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "fullycnnspeechenhancement_amd", "librced_hip.so")
    llvm = "/opt/rocm/lib/llvm/bin/"
    out = {}
    if not os.path.exists(so):
        return out
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so, local)
        subprocess.run([llvm + "llvm-objdump", "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in glob.glob(local + ".*gfx950*"):
            txt = subprocess.run([llvm + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                out[name.split("(")[0].replace("void ", "")] = {k: g(k) for k in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size",
                                                                                 "group_segment_fixed_size")}
    return out


# dynamic LDS of the CR-CED kernels (bytes): the launch's third parameter = Map<FORM>::kLdsBytes of kernels_fused_v3.h; kept here by hand and
# pinned by static_asserts in kernels_fused.hip ("tools/summarize_prof.py prints these")
DYNAMIC_LDS = {"Map<3>": 159024, "Map<2>": 163792, "Map<1>": 162976, "Map<0>": 163024}

WARMUP_DISPATCHES = int(os.environ.get("TRACE_WARMUP", "3"))   # the trace pass's --warmup (tools/profile.sh)


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
    res = code_object_resources()
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        per = defaultdict(list)
        for r in rows:
            n = r.get("Kernel_Name", "")
            if any(s in n for s in ("fused", "final", "conv_layer", "frame16")):
                per[n].append(r)
        print("\n## dispatches (kernel_trace.csv): steady state = without each kernel's first %d dispatches" % WARMUP_DISPATCHES)
        for n, rs in per.items():
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
            steady = d[WARMUP_DISPATCHES:] or d
            r0 = rs[-1]
            full = n.split("(")[0].replace("void ", "")
            co = res.get(full, {})
            lds_dyn = next((v for k, v in DYNAMIC_LDS.items() if k in full), None)
            print("%s" % full)
            print("    calls=%d  avg_all_ns=%.0f  avg_steady_ns=%.0f  min_ns=%d  max_ns=%d" % (len(d), sum(d) / len(d), sum(steady) / len(steady), min(d), max(d)))
            print("    grid=%s workgroup=%s (= %d workgroups)  trace: VGPR_Count=%s SGPR_Count=%s LDS_Block_Size=%s Scratch_Size=%s" % (
                r0.get("Grid_Size_X"), r0.get("Workgroup_Size_X"), int(r0.get("Grid_Size_X", 0)) // max(int(r0.get("Workgroup_Size_X", 1)), 1),
                r0.get("VGPR_Count"), r0.get("SGPR_Count"), r0.get("LDS_Block_Size"), r0.get("Scratch_Size")))
            if co:
                print("    code object: vgpr=%s agpr=%s sgpr=%s scratch=%s static_lds=%s%s" % (
                    co["vgpr_count"], co["agpr_count"], co["sgpr_count"], co["private_segment_fixed_size"], co["group_segment_fixed_size"],
                    "  dynamic_lds=%d B (the launch's; Map<FORM>::kLdsBytes)" % lds_dyn if lds_dyn and "fused_v3" in full else ""))
    print("\n## PMC (average per dispatch)")
    pmc = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for f in find(d, "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = (short(r.get("Kernel_Name", "")), r.get("Counter_Name"))
                    if any(s in k[0] for s in ("fused", "final", "conv_layer", "frame16")):
                        acc[k][0] += float(r.get("Counter_Value", 0) or 0)
                        acc[k][1] += 1
        for (k, c), (tot, n) in sorted(acc.items()):
            print("%-22s %-28s avg=%.6g  (n=%d)" % (k, c, tot / max(n, 1), n))
            pmc[(k, c)] = tot / max(n, 1)
    # derived: how many of the issued MFMAs the nominal FLOPs need, how busy the matrix pipe was
    for k in sorted({k for k, _ in pmc}):
        mf, busy, gui = pmc.get((k, "SQ_INSTS_MFMA")), pmc.get((k, "SQ_VALU_MFMA_BUSY_CYCLES")), pmc.get((k, "GRBM_GUI_ACTIVE"))
        valu = pmc.get((k, "SQ_INSTS_VALU"))
        if mf and k == "fused_v3_kernel":
            frames, flop_frame = 256 * 512, 8207496
            # every product of the default form is six 16x16x32 bf16 MFMAs (16,384 FLOP each): MFMAs the nominal FLOPs need = FLOP / 16384 * 6
            needed = frames * flop_frame / 16384.0 * 6.0
            line = "%-22s derived: mfma_issued=%.4g  mfma_needed(all layers as six bf16 products)=%.4g  padding=%.1f %%" % (k, mf, needed, 100.0 * (1 - needed / mf))
            if valu:
                line += "  other_valu_per_mfma=%.2f" % ((valu - mf) / mf)
            if busy and gui:
                line += "  mfma_pipe_busy=%.1f %% (BUSY_CYCLES / (1024 SIMDs x GUI_ACTIVE / 8))" % (100.0 * busy / (1024.0 * gui / 8.0))
            print(line)
        elif mf and busy and gui:   # any other matrix kernel: VALU mix, pipe occupancy, LDS occupancy
            lds, conf = pmc.get((k, "SQ_LDS_IDX_ACTIVE")), pmc.get((k, "SQ_LDS_BANK_CONFLICT"))
            line = "%-22s derived: mfma_issued=%.4g" % (k, mf)
            if valu:
                line += "  other_valu_per_mfma=%.2f" % ((valu - mf) / mf)
            line += "  mfma_pipe_busy=%.1f %%" % (100.0 * busy / (1024.0 * gui / 8.0))
            if lds:
                line += "  lds_active=%.1f %% of CU-cycles (IDX_ACTIVE / (256 CUs x GUI_ACTIVE / 8)), %.0f %% of it bank conflicts" % (
                    100.0 * lds / (256.0 * gui / 8.0), 100.0 * (conf or 0) / lds)
            print(line)


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2])
