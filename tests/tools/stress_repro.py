"""Race screen: full-batch forward vs slices of it, bit for bit, many times.  usage: stress_repro.py NET dtype N T reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fullycnnspeechenhancement_amd import build_model, weights as _w, spec
net, dtype, N, T, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = build_model(net, False, weights=_w.synthetic_weights(spec.variant_of(net), seed=42), dtype=dtype)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn((N, T, 129, 1), generator=g, device="cuda").abs_()
y = m(x).clone()
bad = 0
rng = np.random.default_rng(0)
for r in range(reps):
    if not torch.equal(m(x), y):
        bad += 1; print("rep", r, "full rerun differs")
    a = int(rng.integers(0, N - 1)); b = int(rng.integers(a + 1, min(N, a + 9) + 1))
    ys = m(x[a:b].contiguous())
    if not torch.equal(ys, y[a:b]):
        d = (ys - y[a:b]).abs(); bad += 1
        print("rep", r, "slice", a, b, "differs: max", float(d.max()), "count", int((d > 0).sum()))
print(net, dtype, N, T, "reps", reps, "mismatches", bad)
