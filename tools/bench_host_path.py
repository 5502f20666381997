#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference's own calling convention (numpy in, numpy out through
rced_forward_host) at BASELINE config 3.  Never the bench `value`; recorded in DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fullycnnspeechenhancement_amd import build_model
from oracle import rced_np
m = build_model("FullyCNNV3", False, weights=rced_np.make_weights("FullyCNNV3"))
x = np.abs(np.random.default_rng(0).standard_normal((256, 512, 129, 1))).astype(np.float32)
m(x[:8])
ts = []
for _ in range(4):
    t0 = time.perf_counter(); y = m(x); ts.append(time.perf_counter() - t0)
print(json.dumps({"host_path_ms": 1e3 * min(ts), "frames_per_s": 256 * 512 / min(ts), "bytes_each_way": x.nbytes}))
