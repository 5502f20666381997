#!/usr/bin/env python3
"""The bf16 R-CED kernel over call sizes (device-resident, ms per call after a warm-up): tools/bench_bf16_shapes.py -> one JSON line.
Used to compare builds (RCED_LIB=exp/<name>.so): the four-wave product against the eight-wave form on small and large calls."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fullycnnspeechenhancement_amd import build_model, weights, spec
out = {}
for net in ("FullyCNNV2", "FullyCNN"):
    m = build_model(net, False, weights=weights.synthetic_weights(spec.variant_of(net)), dtype="bfloat16")
    for (n, t) in ((1, 64), (1, 256), (8, 512), (64, 512), (256, 512)):
        x = torch.randn((n, t, 129, 1), device="cuda").abs_()
        y = torch.empty_like(x)
        for _ in range(150): m(x, out=y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 300
        for _ in range(reps): m(x, out=y)
        torch.cuda.synchronize()
        out["%s %dx%d" % (net, n, t)] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    m.close()
print(json.dumps(out))
