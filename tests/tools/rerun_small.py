"""Rerun small forwards and count bitwise differences.  usage: rerun_small.py NET dtype"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fullycnnspeechenhancement_amd import build_model, weights as _w, spec
net, dtype = sys.argv[1], sys.argv[2]
m = build_model(net, False, weights=_w.synthetic_weights(spec.variant_of(net), seed=42), dtype=dtype)
for (N, T) in ((1, 3), (1, 6), (1, 30), (4, 30), (64, 30), (64, 128)):
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((N, T, 129, 1), generator=g, device="cuda").abs_()
    y = m(x).clone()
    bad = 0; worst = 0.0
    for r in range(20):
        d = (m(x) - y).abs()
        if float(d.max()) > 0: bad += 1; worst = max(worst, float(d.max()))
    print(net, dtype, (N, T), "tiles", N * ((T + 2) // 3), "reruns differing", bad, "/ 20, worst", worst)
