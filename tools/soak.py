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
print(json.dumps(out))
