#!/usr/bin/env python3
"""In-kernel s_memtime breakdown of the bf16 R-CED kernel (kernels_frame16.h), per layer: K loop / wait / epilogue / barrier.
Needs a stamps build:  tools/mkexp.sh st16 "kernels_fused" -DRCED_F16_STAMPS=1 ; RCED_LIB=exp/st16.so python3 tools/stamps16.py [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fullycnnspeechenhancement_amd import build_model, weights as W

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
net = {1: "FullyCNN", 2: "FullyCNNV2"}[variant]
m = build_model(net, False, weights=W.synthetic_weights(variant, seed=42), dtype="bfloat16")
grid = int(os.environ.get("GRID", "0"))
if grid:
    m.set_option("fused_grid", grid)
x = torch.randn((64, 512, 129, 1), device="cuda").abs_()
for _ in range(3):
    y = m(x)
torch.cuda.synchronize()
layers = 9 if variant == 1 else 15
st = [m.get_option("f16stamp%d" % i) for i in range(4 * layers + 1)]
print("layer   kloop    wait  epilog barrier   total")
tot = [0, 0, 0, 0]
for l in range(layers):
    a, b, c, d, e = st[4 * l:4 * l + 5]
    parts = (b - a, c - b, d - c, e - d)
    for i in range(4):
        tot[i] += parts[i]
    print("%5d %7d %7d %7d %7d %7d" % ((l,) + parts + (e - a,)))
print("total %7d %7d %7d %7d %7d" % (tuple(tot) + (sum(tot),)))
print("output layer + its barrier %d; tile %d" % (st[-1] - st[-2], st[-1] - st[0]))
