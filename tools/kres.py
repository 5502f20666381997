#!/usr/bin/env python3
"""Register / scratch / LDS usage of the compiled gfx950 kernels (from the code objects' amdhsa metadata).
Usage: tools/kres.py [pattern] [object or .so ...]   (default: build/train_api.o build/kernels_fused.o build/train_mfma_v2.o)"""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
objs = sys.argv[2:] or [os.path.join(ROOT, "build", n) for n in ("train_api.o", "kernels_fused.o", "train_mfma_v2.o")]
with tempfile.TemporaryDirectory() as tmp:
    for o in objs:
        local = os.path.join(tmp, os.path.basename(o))
        shutil.copy(o, local)
        subprocess.run([LLVM + "llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in glob.glob(local + ".*gfx950*"):
            txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
                name = g("name")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                short = dem.split("(")[0].replace("void ", "").replace("rced::", "")
                if pat in short:
                    print("%-70s vgpr %4s agpr %4s sgpr %4s scratch %5s lds %6s" % (short[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
                                                                                       g("private_segment_fixed_size"), g("group_segment_fixed_size")))
