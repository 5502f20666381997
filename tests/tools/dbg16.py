#!/usr/bin/env python3
"""Which skip does the bf16 kernel get wrong?  Compares the GPU result with the emulation run with one skip source dropped / zeroed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import rced_np, layers as L
from fullycnnspeechenhancement_amd import build_model

net = sys.argv[1] if len(sys.argv) > 1 else "FullyCNNV2"
w = rced_np.make_weights(net, seed=42)
x = rced_np.make_input(3, 20, seed=5)
m = build_model(net, False, weights=w, dtype="bfloat16")
y = m(x)
ref = rced_np.forward_bf16(net, w, x)
sc = np.abs(ref).max()
d = np.abs(y - ref)[..., 0]
print("full: %.3e" % (d.max() / sc))
print("per frame max err (utt 0):", np.array2string(d[0].max(axis=1) / sc, precision=3))
print("per bin max err:", np.array2string(d.max(axis=(0, 1))[::8] / sc, precision=3))
lay = L.layers_for(net)
orig = [l.skip_pre for l in lay]
for i, l in enumerate(lay):
    if orig[i] < 0:
        continue
    lay2 = [ll._replace(skip_pre=-1) if k == i else ll for k, ll in enumerate(lay)]
    L_layers_for = L.layers_for
    L.layers_for = lambda n, lay2=lay2: lay2
    try:
        r = rced_np.forward_bf16(net, w, x)
    finally:
        L.layers_for = L_layers_for
    print("emulation without the skip into layer %d (%s): err %.3e" % (i, l.scope, np.abs(y - r).max() / sc))

# hypothesis: the skip into layer `LH` is missing for the first two tile groups (bins < 96) only
def emu_partial(LH, lo_bins):
    layers = L.layers_for(net)
    tensors = [rced_np.bf16_round(np.asarray(x, np.float32)).astype(np.float64)]
    for i, l in enumerate(layers):
        k = np.asarray(w[l.scope + "/kernel"], np.float64)
        shift = np.asarray(w[l.scope + "/bias"], np.float64)
        if l.use_norm:
            p = l.scope + "/batch_norm/"
            g, b, mm, v = (np.asarray(w[p + nn], np.float64) for nn in ("gamma", "beta", "moving_mean", "moving_variance"))
            s = g / np.sqrt(v + L.BN_EPS)
            k = k * s
            shift = (shift - mm) * s + b
        k = rced_np.bf16_round(k.astype(np.float32))
        yy = rced_np.conv2d_same(tensors[l.src], k, shift.astype(np.float32), np.float64)
        if l.skip_pre >= 0:
            sk = tensors[l.skip_pre].copy()
            if i == LH:
                sk[:, :, :lo_bins] = 0
            yy = yy + sk
        if l.use_act:
            yy = np.maximum(yy, 0)
        if i != len(layers) - 1:
            yy = rced_np.bf16_round(yy.astype(np.float32)).astype(np.float64)
        tensors.append(yy)
    return tensors[-1].astype(np.float32)

for LH in (11, 12, 13, 14):
    for lo in (48, 96):
        r = emu_partial(LH, lo)
        print("emulation with the skip into layer %d zero below bin %d: err %.3e" % (LH, lo, np.abs(y - r).max() / sc))
