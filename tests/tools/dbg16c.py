#!/usr/bin/env python3
"""bf16 R-CED kernel, several tiles per workgroup (fused_grid small): per-frame error against the emulation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import rced_np
from fullycnnspeechenhancement_amd import build_model
net = sys.argv[1] if len(sys.argv) > 1 else "FullyCNNV2"
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = rced_np.make_weights(net, seed=42)
x = rced_np.make_input(1, 64, seed=5)
m = build_model(net, False, weights=w, dtype="bfloat16")
m.set_option("fused_grid", grid)
y = m(x)
ref = rced_np.forward_bf16(net, w, x)
sc = np.abs(ref).max()
d = np.abs(y - ref)[0, :, :, 0] / sc
print("grid %d: max err %.3e" % (grid, d.max()))
print("per frame:", " ".join("%.0e" % v for v in d.max(axis=1)))
bad = np.argwhere(d > 1e-2)
print("bad (frame, bin) count", len(bad), "first", bad[:12].tolist())
