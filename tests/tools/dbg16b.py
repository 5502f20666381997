#!/usr/bin/env python3
"""Difference between two builds of the library on the bf16 R-CED forward: RCED_LIB_A / RCED_LIB_B (each in a child process)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from oracle import rced_np
    from fullycnnspeechenhancement_amd import build_model
    net = "FullyCNNV2"
    w = rced_np.make_weights(net, seed=42)
    x = rced_np.make_input(1, 8, seed=5)
    ch = int(os.environ.get("DELTA_CH", "-1"))
    if ch >= 0:   # the output layer as a delta: the mask IS channel ch of the last fused layer's output
        k = np.zeros_like(w["decode_8/kernel"])
        k[0, 64, ch, 0] = 1.0
        w["decode_8/kernel"] = k
        w["decode_8/bias"] = np.zeros_like(w["decode_8/bias"])
    m = build_model(net, False, weights=w, dtype="bfloat16")
    np.save(sys.argv[2], m(x))
    sys.exit(0)
outs = []
for k in ("A", "B"):
    env = dict(os.environ, RCED_LIB=os.environ["RCED_LIB_" + k])
    f = "/tmp/dbg16b_%s.npy" % k
    subprocess.check_call([sys.executable, __file__, "child", f], env=env)
    outs.append(np.load(f)[0, :, :, 0])
a, b = outs
d = np.abs(a - b) / np.abs(b).max()
np.set_printoptions(linewidth=250, precision=1, suppress=False)
for t in range(d.shape[0]):
    print("frame %d:" % t, " ".join("%d:%.0e" % (i, v) for i, v in enumerate(d[t]) if v > 1e-6))
