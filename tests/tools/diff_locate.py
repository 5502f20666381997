"""Where do two runs of the same forward differ?  usage: diff_locate.py NET dtype N T"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fullycnnspeechenhancement_amd import build_model, weights as _w, spec
net, dtype, N, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
m = build_model(net, False, weights=_w.synthetic_weights(spec.variant_of(net), seed=42), dtype=dtype)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn((N, T, 129, 1), generator=g, device="cuda").abs_()
y0 = m(x).clone().cpu().numpy()[..., 0]
bins = collections.Counter(); frames = collections.Counter(); utts = collections.Counter()
for r in range(6):
    y = m(x).cpu().numpy()[..., 0]
    n, t, f = np.nonzero(y != y0)
    bins.update(f.tolist()); frames.update((t % 3).tolist()); utts.update(n.tolist())
    if r == 0 and len(n):
        # blobs: list the (utt, frame) pairs and their bin ranges
        seen = {}
        for a, b, c in zip(n, t, f): seen.setdefault((int(a), int(b)), []).append(int(c))
        for k in list(seen)[:12]: print("utt/frame", k, "bins", min(seen[k]), "..", max(seen[k]), "count", len(seen[k]))
print("frame-in-tile histogram", dict(frames))
print("bin histogram (top)", sorted(bins.items(), key=lambda kv: -kv[1])[:20])
print("bins min/max", min(bins) if bins else None, max(bins) if bins else None, "distinct utts", len(utts))
