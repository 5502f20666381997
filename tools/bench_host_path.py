#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference's own calling convention (numpy in, numpy out through
rced_forward_host) at BASELINE config 3, for several pipeline depths (option "host_chunks"; 1 = copy in,
run, copy out).  Never the bench `value`; recorded in DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fullycnnspeechenhancement_amd import build_model
from fullycnnspeechenhancement_amd import weights as _weights
m = build_model("FullyCNNV3", False, weights=_weights.synthetic_weights(3, seed=42))
x = np.abs(np.random.default_rng(0).standard_normal((256, 512, 129, 1))).astype(np.float32)
m(x[:8])
ref = m(torch.from_numpy(x).cuda()).cpu().numpy()
out = {"bytes_each_way": x.nbytes}
for chunks in (1, 2, 4, 8, 16, 32):
    m.set_option("host_chunks", chunks)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); y = m(x); ts.append(time.perf_counter() - t0)
    out["chunks_%d" % chunks] = {"ms": round(1e3 * min(ts), 3), "frames_per_s": round(256 * 512 / min(ts)),
                                 "identical_to_resident": bool(np.array_equal(y, ref))}
print(json.dumps(out))
# reference points on this box: raw copies of the same array through torch (pageable and pinned)
xt = torch.from_numpy(x)
d = torch.empty_like(xt, device="cuda")
def tm(f, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(1e3 * min(ts), 3)
xp = xt.pin_memory()
yp = torch.empty_like(xp).pin_memory()
print(json.dumps({"h2d_pageable_ms": tm(lambda: d.copy_(xt)), "h2d_pinned_ms": tm(lambda: d.copy_(xp)),
                  "d2h_pinned_ms": tm(lambda: yp.copy_(d)), "d2h_pageable_ms": tm(lambda: d.cpu()),
                  "np_empty_touch_ms": tm(lambda: np.empty_like(x).fill(0)), "cpus": os.cpu_count()}))
