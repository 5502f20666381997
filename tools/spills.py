#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions sit: address, kind, and the nearest loop (backward branch) around them.
Usage: tools/spills.py <object> <mangled-name-substring>"""
import glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
obj, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    local = os.path.join(tmp, os.path.basename(obj)); shutil.copy(obj, local)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for co in glob.glob(local + ".*gfx950*"):
        txt = subprocess.run([LLVM + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        fn, ins = None, {}
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m: fn = m.group(1); ins[fn] = []; continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m and fn: ins[fn].append((int(m.group(3), 16), m.group(1), m.group(2)))
        for fn, L in ins.items():
            if pat not in fn: continue
            print(fn[:120])
            loops = []
            for a, op, args in L:
                if op.startswith(("s_cbranch", "s_branch")):
                    off = int(args.split()[-1]);  off = off - 65536 if off > 32767 else off
                    tgt = a + 4 + 4 * off
                    if tgt < a: loops.append((tgt, a))
            for a, op, args in L:
                if op.startswith("scratch_"):
                    inl = [l for l in loops if l[0] <= a <= l[1]]
                    inner = min(inl, key=lambda l: l[1] - l[0]) if inl else None
                    print("  %x %-22s %-30s %s" % (a, op, args[:30], ("in loop %x..%x (%d B)" % (inner[0], inner[1], inner[1] - inner[0])) if inner else "outside loops"))
            n = sum(1 for a, op, args in L if op.startswith("v_mfma"))
            print("  mfma instrs: %d, total %d" % (n, len(L)))
