#!/usr/bin/env python3
"""Lint of the compiled gfx950 kernels (the shipped librced_hip.so by default) for three instruction sequences that hipcc emits without complaint and
that misbehaved on MI355X in this project (the reproducibility hunts of rounds 2 and 6):

  A. VALU writes VCC  /  vector-memory instruction  /  SALU reads VCC   (three consecutive instructions).
     Measured: `v_cmp_gt_i32 vcc` / `buffer_store_dwordx4` / `s_and_saveexec_b64 s[0:1], vcc` lost lanes of the LDS store
     the mask guards (run-to-run differences, two workgroups per CU); one wait state between store and SALU cures it.
     Also the same shape through an SGPR pair instead of VCC: `v_cmp_*_e64 s[a:b]` / vector-memory / SALU reads s[a:b]
     (s_and_saveexec_b64, s_mov_b64 exec, ...) -- which register the compare lands in is the allocator's choice.
  B. An LDS-DMA with an SGPR base (`global_load_lds_* v, s[a:b]`, issued from inline asm, where hipcc pads no hazard wait
     states) fewer than 5 wait states behind a VALU instruction that wrote s[a] or s[b] (v_readlane of a spilled SGPR,
     v_readfirstlane).

  C. A matrix instruction of one shape DIRECTLY behind a matrix instruction of ANOTHER shape whose destination it reads as srcC
     (`v_mfma_f32_16x16x32_bf16 v[a:b], ...` / `v_mfma_f32_16x16x16_bf16 v[a:b], ..., v[a:b]`).  The same-shape chain is what the
     hardware interlocks; the mixed pair read the accumulator before the first instruction's last pass had written it (round 6,
     kernels_frame16.h: one tile's skip added to a stale sum, deterministic) and hipcc pads nothing between them.

Usage: tools/isa_lint.py [object or shared-object files...]   (default: the library the package loads,
fullycnnspeechenhancement_amd/librced_hip.so -- the code that actually runs, not objects that may be stale or absent).  Exit status 1 and a
listing when a sequence is found.  tests/test_host.py runs it on every build."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

VALU_WRITES_VCC = re.compile(r"^v_cmpx?_\w+_e32\b|^v_\w+\s+(v\d+|v\[\d+:\d+\]), vcc\b|^v_cmpx?_\w+\s+vcc\b")
VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)")
SALU_READS_VCC = re.compile(r"^s_\w+\s+[^,]+,.*\bvcc\b|^s_cbranch_vcc")
VALU_WRITES_SPAIR = re.compile(r"^v_cmpx?_\w+\s+s\[(\d+):(\d+)\]")
SALU_READS_SPAIR = re.compile(r"^s_\w+\s+[^,]+,.*\bs\[(\d+):(\d+)\]")
DMA_SBASE = re.compile(r"^global_load_lds_\w+\s+v\d+, s\[(\d+):(\d+)\]")
VALU_WRITES_SGPR = re.compile(r"^v_(readlane|readfirstlane)_b32\s+s(\d+)\b")
NOP = re.compile(r"^s_nop\s+(\d+)")
MFMA = re.compile(r"^(v_mfma_\w+)\s+(v\[\d+:\d+\]|a\[\d+:\d+\]), (\S+), (\S+), (v\[\d+:\d+\]|a\[\d+:\d+\]|\S+)")


def disassemble(obj, tmp):
    """[(function, [instruction text...])] of the gfx950 code object embedded in a host object."""
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cos = [p for p in glob.glob(local + ".*") if "gfx950" in p]
    out = []
    for co in cos:
        txt = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
        fn, ins = None, []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if fn is not None:
                    out.append((fn, ins))
                fn, ins = m.group(1), []
                continue
            m = re.match(r"^\s+(\S.*?)\s*//", line)      # "   v_add_u32 ...   // 0000...: ENCODING"
            if m and fn is not None:
                ins.append(m.group(1).strip())
        if fn is not None:
            out.append((fn, ins))
    return out


def wait_states(ins):
    m = NOP.match(ins)
    return int(m.group(1)) + 1 if m else 1


def lint_function(fn, ins):
    found = []
    for i in range(len(ins) - 2):
        if VALU_WRITES_VCC.search(ins[i]) and VMEM.search(ins[i + 1]) and SALU_READS_VCC.search(ins[i + 2]):
            found.append(("A", fn, i, ins[i:i + 3]))
            continue
        w = VALU_WRITES_SPAIR.match(ins[i])
        if w and VMEM.search(ins[i + 1]):
            r = SALU_READS_SPAIR.match(ins[i + 2])
            # the SALU instruction's SOURCE operands: everything behind the first comma
            if r and ("s[%s:%s]" % w.groups()) in ins[i + 2].split(",", 1)[1]:
                found.append(("A", fn, i, ins[i:i + 3]))
    for i in range(len(ins) - 1):
        m0, m1 = MFMA.match(ins[i]), MFMA.match(ins[i + 1])
        if m0 and m1 and m0.group(1) != m1.group(1) and m1.group(5).rstrip(",") == m0.group(2):
            found.append(("C", fn, i, ins[i:i + 2]))
    for i, s in enumerate(ins):
        m = DMA_SBASE.match(s)
        if not m:
            continue
        regs = {int(m.group(1)), int(m.group(2))}
        ws = 0
        for j in range(i - 1, max(i - 8, -1), -1):
            w = VALU_WRITES_SGPR.match(ins[j])
            if w and int(w.group(2)) in regs and ws < 5:
                found.append(("B", fn, j, ins[j:i + 1]))
                break
            ws += wait_states(ins[j])
            if ws >= 5:
                break
    return found


def lint(objs):
    found = []
    nfn = nins = 0
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            for fn, ins in disassemble(o, tmp):
                nfn += 1
                nins += len(ins)
                found += lint_function(fn, ins)
    return found, nfn, nins


def default_objects():
    so = os.path.join(ROOT, "fullycnnspeechenhancement_amd", "librced_hip.so")
    return [so] if os.path.exists(so) else []


def main():
    objs = sys.argv[1:] or default_objects()
    if not objs:
        print("isa_lint: no library (run __graft_entry__.build() first)")
        return 2
    found, nfn, nins = lint(objs)
    for kind, fn, i, seq in found:
        print("sequence %s in %s at instruction %d:\n    %s" % (kind, fn[:100], i, "\n    ".join(seq)))
    print("isa_lint: %d kernels, %d instructions, %d finding(s)" % (nfn, nins, len(found)))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
