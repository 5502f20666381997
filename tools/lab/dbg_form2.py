"""Diagnostic: the fused form (v3_l2x6 = 2) against the X6 form on one random input: where (frame mod 4, bin) do they differ?"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import rced_np
from fullycnnspeechenhancement_amd import model as M
w = rced_np.make_weights("FullyCNNV3", seed=7)
x = rced_np.make_input(1, 16, seed=3)
def run(form):
    m = M.FullyCNNSEModelV3(False, weights=w, device=0)
    m.set_option("v3_l2x6", form)
    return np.asarray(m(x), dtype=np.float64).reshape(16, 129)
a, b = run(1), run(2)
sc = np.abs(a).max()
d = np.abs(a - b) / sc
print("max err", d.max(), "nan", np.isnan(b).sum())
np.set_printoptions(linewidth=250, precision=1, suppress=False)
for f in range(8):
    print("frame", f, " ".join("%.0e" % v if v > 1e-5 else "." for v in d[f]))
