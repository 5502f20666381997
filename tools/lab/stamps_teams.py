#!/usr/bin/env python3
"""Diagnostic: RCED_STAMPS build, two-team kernel (RCED_V3_TEAMS=1): per wave of workgroup 0, kilo-cycles in
enter-wait / L1 / L2 / L3 math / leave (incl. refill) / team barrier / other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RCED_LIB", os.path.join(ROOT, "exp", "librced_stamps.so"))
os.environ["RCED_V3_TEAMS"] = "1"
sys.path.insert(0, ROOT)
import torch
from fullycnnspeechenhancement_amd import build_model
from oracle import rced_np
m = build_model("FullyCNNV3", False, weights=rced_np.make_weights("FullyCNNV3"))
x = torch.randn((256, 512, 129, 1), device="cuda").abs_()
y = m(x)
torch.cuda.synchronize()
print("wave team role | enter   L1     L2     L3   leave  tbar   other | total (kilo-cycles)")
for w in range(8):
    v = [m.get_option("stamp%d" % (w * 8 + i)) for i in range(8)]
    team = w >> 2
    print("%3d %4d %4d | %6d %6d %6d %6d %6d %6d %6d | %d" % (w, team, (w + 2 * team) & 3, v[0], v[1], v[2], v[3], v[4], v[5], v[6] + v[7], sum(v)))
