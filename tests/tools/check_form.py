#!/usr/bin/env python3
"""Quick look: the CR-CED goldens and a ragged random batch through every form of the fused kernel (v3_l2x6 = 0..3)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_err
from fullycnnspeechenhancement_amd import build_model
from oracle import rced_np, rced_c
w, g = load_golden("v3")
forms = [int(a) for a in sys.argv[1:]] or [3, 2, 0]
for form in forms:
    m = build_model("FullyCNNV3", False, weights=w)
    m.set_option("v3_l2x6", form)
    for key in ("small", "long", "c1"):
        y = m(g["x_" + key])
        print("form %d %-6s rel err %.3e  finite %s" % (form, key, rel_err(y, g["y_" + key]), np.isfinite(y).all()), flush=True)
    x = rced_np.make_input(3, 37, seed=77)
    ref = rced_c.forward("FullyCNNV3", w, x, np.float64)
    y = m(x)
    e = np.abs(y.astype(np.float64) - ref).reshape(3, 37, 129)
    print("form %d ragged 3x37: rel err %.3e; worst frame %s worst bin %s" % (form, rel_err(y, ref), np.unravel_index(e.max(axis=2).argmax(), (3, 37)), e.max(axis=(0, 1)).argmax()), flush=True)
    m.close()
