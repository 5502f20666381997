#!/usr/bin/env python3
"""Soak: many launches of every fused kernel at config-3 size must give bit-identical outputs (the kernels hand partial
sums between waves through LDS flags; a race would show up as a rare different bit)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fullycnnspeechenhancement_amd import build_model, spec, weights
out = {}
for net, dtype, reps in (("FullyCNNV3", "float32", 400), ("FullyCNN", "float32", 150), ("FullyCNNV2", "float32", 150),
                         ("FullyCNN", "bfloat16", 150), ("FullyCNNV2", "bfloat16", 150)):
    m = build_model(net, False, weights=weights.synthetic_weights(spec.variant_of(net)), dtype=dtype)
    x = torch.randn((256, 512, 129, 1), device="cuda").abs_()
    ref = m(x).clone()
    bad = 0
    for i in range(reps):
        y = m(x)
        if not torch.equal(y, ref):
            bad += 1
    torch.cuda.synchronize()
    out["%s %s" % (net, dtype)] = {"launches": reps, "different": bad, "finite": bool(torch.isfinite(ref).all())}
# the R-CED kernels' latency form (one-frame tiles; calls with fewer 3-frame tiles than CUs): BASELINE config 1's shape, many launches
for net in ("FullyCNN", "FullyCNNV2"):
    m = build_model(net, False, weights=weights.synthetic_weights(spec.variant_of(net)))
    x = torch.randn((1, 256, 129, 1), device="cuda").abs_()
    ref = m(x).clone()
    bad = sum(0 if torch.equal(m(x), ref) else 1 for _ in range(1000))
    m.set_option("latency_form", 0)
    same = bool(torch.equal(m(x), ref))
    out["%s float32 [1,256] latency form" % net] = {"launches": 1000, "different": bad, "equals_3_frame_form": same}
print(json.dumps(out))
