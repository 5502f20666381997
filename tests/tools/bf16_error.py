import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from fullycnnspeechenhancement_amd import build_model
from oracle import rced_np
for net in ("FullyCNN", "FullyCNNV2"):
    w = rced_np.make_weights(net, seed=42); x = rced_np.make_input(4, 64, seed=9)
    m = build_model(net, False, weights=w, dtype="bfloat16")
    y = m(x); r16 = rced_np.forward_bf16(net, w, x); r32 = rced_np.forward(net, w, x)
    d = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print(net, "vs emulation %.2e  vs fp32 oracle %.2e  emulation vs fp32 %.2e  wgs/cu=%s" % (d(y, r16), d(y, r32), d(r16, r32), "?"))
