import ctypes, os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fullycnnspeechenhancement_amd import build_model, _lib
from fullycnnspeechenhancement_amd import weights as _weights
m = build_model("FullyCNNV3", False, weights=_weights.synthetic_weights(3, seed=42))
lib = _lib.load()
x = np.abs(np.random.default_rng(0).standard_normal((256, 512, 129, 1))).astype(np.float32)
y = np.zeros_like(x)
xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
def tm(f, n=4):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(1e3 * min(ts), 2)
res = {}
res["device_forward_ms"] = tm(lambda: m(xd))
for ch in (1, 8):
    m.set_option("host_chunks", ch)
    res["host_reused_y_ch%d" % ch] = tm(lambda: _lib.check(lib.rced_forward_host(m._handle, x.ctypes.data, y.ctypes.data, 256, 512)))
    res["host_fresh_y_ch%d" % ch] = tm(lambda: m(x))
xp = torch.from_numpy(x).pin_memory(); yp = torch.empty_like(xp).pin_memory()
for ch in (1, 8):
    m.set_option("host_chunks", ch)
    res["host_pinned_ch%d" % ch] = tm(lambda: _lib.check(lib.rced_forward_host(m._handle, xp.data_ptr(), yp.data_ptr(), 256, 512)))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
res["raw_hipMemcpy_h2d_pageable"] = tm(lambda: hip.hipMemcpy(xd.data_ptr(), x.ctypes.data, x.nbytes, 1))
res["raw_hipMemcpy_d2h_pageable"] = tm(lambda: hip.hipMemcpy(y.ctypes.data, xd.data_ptr(), x.nbytes, 2))
res["raw_hipMemcpy_h2d_pinned"] = tm(lambda: hip.hipMemcpy(xd.data_ptr(), xp.data_ptr(), x.nbytes, 1))
print(json.dumps(res))
