"""Diagnostic: per-frame error of the fused form against the fp32-MFMA form and the C oracle."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import rced_np, rced_c
from fullycnnspeechenhancement_amd import model as M
T = int(os.environ.get("T", "16")); N = int(os.environ.get("N", "2"))
w = rced_np.make_weights("FullyCNNV3", seed=7)
x = rced_np.make_input(N, T, seed=3)
mode = os.environ.get("MODE", "")
for blk in (("CE1", "CE2", "CE3", "CD1", "CD2") if mode.startswith("all") else ("CE2", "CE3", "CD1", "CD2")):
    pre = blk + "_encode_1"
    if mode == "norem":      # channels 16, 17 of the 8 -> 18 layers output zero: the remainder pass contributes nothing
        w[pre + "/kernel"][..., 16:] = 0; w[pre + "/bias"][16:] = 0; w[pre + "/batch_norm/beta"][16:] = 0; w[pre + "/batch_norm/moving_mean"][16:] = 0
    if mode == "nomain":     # channels 0..15 output zero
        w[pre + "/kernel"][..., :16] = 0; w[pre + "/bias"][:16] = 0; w[pre + "/batch_norm/beta"][:16] = 0; w[pre + "/batch_norm/moving_mean"][:16] = 0
    if mode == "zero":       # no weights at all: the layer's output is relu(shift)
        w[pre + "/kernel"][...] = 0
    if mode == "allzero":    # every block's first layer outputs relu(shift): the net is a constant
        w[pre + "/kernel"][...] = 0
    if mode == "allzero2":   # ... and every block's second layer too
        w[pre + "/kernel"][...] = 0; w[blk + "_encode_2/kernel"][...] = 0
    if mode == "allzero3":   # ... and every block's third layer: the net is decode_final of constants
        w[pre + "/kernel"][...] = 0; w[blk + "_encode_2/kernel"][...] = 0; w[blk + "_decode/kernel"][...] = 0
    if mode == "zero0":      # no weights, no shift: the layer's output is 0
        w[pre + "/kernel"][...] = 0; w[pre + "/bias"][...] = 0; w[pre + "/batch_norm/beta"][...] = 0; w[pre + "/batch_norm/moving_mean"][...] = 0
    if mode == "ch0":        # only input channel 0, centre tap
        k = w[pre + "/kernel"]; k[:, :4] = 0; k[:, 5:] = 0; k[:, :, 1:] = 0
    if mode == "tap0":       # only tap 0
        k = w[pre + "/kernel"]; k[:, 1:] = 0
    if mode == "tap4":       # only the centre tap of the 8 -> 18 layers
        k = w[pre + "/kernel"]; k[:, :4] = 0; k[:, 5:] = 0
ref = rced_c.forward("FullyCNNV3", w, x, np.float64).reshape(N, T, 129)
def run(form):
    m = M.FullyCNNSEModelV3(False, weights=w, device=0)
    m.set_option("v3_l2x6", form)
    return np.asarray(m(x), dtype=np.float64).reshape(N, T, 129)
if mode.startswith("z14"):
    for b_ in ("CE2", "CE3", "CD1", "CD2"):
        for n_ in ("_encode_1", "_encode_2", "_decode"): w[b_ + n_ + "/kernel"][...] = 0
    if mode == "z14l1": w["CE1_encode_1/kernel"][...] = 0
    if mode == "z14l2": w["CE1_encode_2/kernel"][...] = 0
    if mode == "z14l3": w["CE1_decode/kernel"][...] = 0
if mode == "b0zero":
    for n_ in ("CE1_encode_1", "CE1_encode_2", "CE1_decode"): w[n_ + "/kernel"][...] = 0
if mode == "b0l1zero": w["CE1_encode_1/kernel"][...] = 0
if mode == "b0l2zero": w["CE1_encode_2/kernel"][...] = 0
if mode == "b0l3zero": w["CE1_decode/kernel"][...] = 0
ref = rced_c.forward("FullyCNNV3", w, x, np.float64).reshape(N, T, 129)
sc = np.abs(ref).max()
for form in (0, 2):
    y = run(form)
    d = np.abs(y - ref) / sc
    print("form", form, "max err", d.max())
    for n in range(N):
        print("  utt", n, " ".join("%.0e" % v for v in d[n].max(axis=1)))
    if form == 2 and os.environ.get("BINS"):
        print("  frame 5 by bin:", " ".join("%.0e" % v if v > 1e-5 else "." for v in d[0, 5]))
        print("  y   :", " ".join("%.2f" % v for v in y[0, 5, :24]))
        print("  ref :", " ".join("%.2f" % v for v in ref[0, 5, :24]))
