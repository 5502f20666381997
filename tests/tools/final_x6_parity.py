"""The output layer on the bf16 matrix pipe (kernels_final_x6.h) against the golden vectors: run with RCED_FINAL_X6=0 / 1 to compare
the three-part form with the fp32 MFMA kernel (tests use the default, the three-part form)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))   # run from the repo root
from conftest import NETS, load_golden
from fullycnnspeechenhancement_amd import build_model
for net_work, tag, variant in NETS:
    w, g = load_golden(tag)
    m = build_model(net_work, False, weights=w)
    x = g["x"] if "x" in g else None
    keys = list(g.keys())
    xs = [k for k in keys if k.startswith("x")]
    for kx in xs:
        ky = kx.replace("x", "y", 1)
        if ky not in g: continue
        y = m(np.ascontiguousarray(g[kx]))
        ref = g[ky]
        print(net_work, kx, "max rel err %.3e  (x6=%s)" % (np.abs(y - ref).max() / np.abs(ref).max(), os.environ.get("RCED_FINAL_X6", "1")))
