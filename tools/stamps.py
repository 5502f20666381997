#!/usr/bin/env python3
"""Diagnostic: run one forward with the RCED_STAMPS build (exp/librced_stamps.so) and print, for
each wave of workgroup 0, kilo-cycles spent in each layer kind's math and barrier wait."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RCED_LIB", os.path.join(ROOT, "exp", "stamps.so"))
sys.path.insert(0, ROOT)
import torch
from fullycnnspeechenhancement_amd import build_model
from fullycnnspeechenhancement_amd import weights as _weights
m = build_model("FullyCNNV3", False, weights=_weights.synthetic_weights(3, seed=42))
if os.environ.get("V3_L2X6") is not None:   # 0: the F32 form of the kernel (every layer on the fp32 MFMA)
    m.set_option("v3_l2x6", int(os.environ["V3_L2X6"]))
x = torch.randn((256, 512, 129, 1), device="cuda").abs_()
y = m(x)
torch.cuda.synchronize()
print("wave   L1math  L2math  L3math | L1wait  L2wait  L3wait | L1first math/wait   (kilo-cycles)")
for w in range(8):
    v = [m.get_option("stamp%d" % (w * 8 + i)) for i in range(8)]
    print("%4d  %7d %7d %7d | %7d %7d %7d | %7d %7d   total %d" % (w, *v, sum(v)))
print("decode_final phase (kilo-cycles) per wave:", [m.get_option("stamp%d" % (64 + w * 3)) for w in range(8)])
print("detail (kilo-cycles; the stamps themselves cost ~100 cycles each and serialise the wave): "
      "L3 share / regular job / collect / epilogue | L1 remainder / single / pairs | L2 share")
for w in range(8):
    v = [m.get_option("stamp%d" % (88 + w * 16 + i)) for i in range(8)]
    print("%4d  %7d %7d %7d %7d | %7d %7d %7d | %7d" % (w, *v))
print("slots 8..15 (all-x6 form: decode_final's GEMM / barrier / reduce / convert / barrier; RCED_STAMPS=2 builds: the bf16-pipe layer 1's pair job, slot by slot (six slots, then the last pair's epilogue), kilo-cycles over all tiles")
for w in range(8):
    v = [m.get_option("stamp%d" % (88 + w * 16 + 8 + i)) for i in range(8)]
    print("%4d  " % w + " ".join("%7d" % x for x in v))
