import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fullycnnspeechenhancement_amd import build_model, weights, spec
out = {}
for net in ("FullyCNN", "FullyCNNV2", "FullyCNNV3"):
    m = build_model(net, False, weights=weights.synthetic_weights(spec.variant_of(net)))
    for (n, t) in ((1, 256), (1, 64), (8, 512)):
        x = torch.randn((n, t, 129, 1), device="cuda").abs_()
        for _ in range(5): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): m(x)
        torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / 50
        xn = x.cpu().numpy()
        for _ in range(3): m(xn)
        t0 = time.perf_counter()
        for _ in range(20): m(xn)
        host = (time.perf_counter() - t0) / 20
        out["%s %dx%d" % (net, n, t)] = {"device_us": round(dev * 1e6, 1), "numpy_us": round(host * 1e6, 1)}
print(json.dumps(out, indent=1))
