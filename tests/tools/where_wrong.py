"""Where does a fused forward differ from the oracle?  usage: where_wrong.py NET N T"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fullycnnspeechenhancement_amd import build_model
from oracle import rced_np
net, n, t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = rced_np.make_weights(net)
x = np.abs(np.random.default_rng(5).standard_normal((n, t, 129, 1))).astype(np.float32)
m = build_model(net, False, weights=w)
y = m(x)
ref = rced_np.forward(net, w, x)
d = np.abs(y - ref)[..., 0]
print("max err", d.max(), "scale", np.abs(ref).max())
for i in range(n):
    for f in range(t):
        bad = np.nonzero(d[i, f] > 1e-4 * np.abs(ref).max())[0]
        if len(bad):
            print("utt", i, "frame", f, "bad bins", bad.min(), "..", bad.max(), "count", len(bad), "max", d[i, f].max())
