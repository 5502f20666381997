#!/usr/bin/env python3
"""Fuzz of the bf16 R-CED kernel (kernels_frame16.h), outside the test suite (GPU box): random weights, shapes and input scales for
R-CED V1 / V2 with option bf16, each against the numpy emulation with the same rounding places (oracle/rced_np.forward_bf16), bit-identical
over grids of 1 / 3 / the default number of workgroups, over both forms of the kernel (four / eight frames per workgroup) and over two calls.  Criterion: TWICE the suite's bf16 bounds (tests/test_forward_gpu.py
holds its fixed seeds to 1e-2 of the scale element-wise, 1e-3 rms) or three times the emulation's own accumulation noise on that case --
the distance between its fp64- and fp32-accumulating runs, which reaches 2e-2 / 1.5e-3 where a small input leaves the shifts in charge
(a sum that lands on a bf16 midpoint rounds either way, and fifteen layers pass the step on): the kernel's fp32 sums are a third order of
summation, no closer to either.  A kernel defect shows as a multiple of that; the cases over the suite's own bounds are counted.  Usage: python tests/tools/fuzz_bf16.py [cases per net, default 60] -> one JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fullycnnspeechenhancement_amd import build_model
from oracle import rced_np

EL, RMS = 1e-2, 1e-3
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = {}
t_start = time.time()
for net in ("FullyCNN", "FullyCNNV2"):
    rng = np.random.default_rng({"FullyCNN": 1101, "FullyCNNV2": 2202}[net])
    worst_el = worst_rms = noise_el = noise_rms = 0.0
    over = 0
    shapes, failed = [], []
    model = None
    for i in range(cases):
        if i % 6 == 0:           # new weights every six cases
            if model is not None:
                model.close()
            wseed = int(rng.integers(1, 1 << 30))
            w = rced_np.make_weights(net, seed=wseed)
            model = build_model(net, False, weights=w, dtype="bfloat16")
        n = int(rng.choice([1, 1, 2, 3, 5, 9]))
        t = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 64, 100, 129]))
        if n * t > 600:
            t = max(1, 600 // n)
        scale = float(10.0 ** rng.uniform(-2, 2))
        x = (rced_np.make_input(n, t, seed=int(rng.integers(1, 1 << 30))) * scale).astype(np.float32)
        ref = rced_np.forward_bf16(net, w, x)
        alt = rced_np.forward_bf16(net, w, x, accumulate=np.float32).astype(np.float64)
        model.set_option("fused_grid", 0)
        y = model(x)
        den = max(float(np.abs(ref).max()), 1e-30)
        el = float(np.abs(y.astype(np.float64) - ref).max() / den)
        rms = float(np.sqrt(np.mean((y.astype(np.float64) - ref) ** 2)) / den)
        n_el, n_rms = float(np.abs(alt - ref).max() / den), float(np.sqrt(np.mean((alt - ref) ** 2)) / den)
        noise_el, noise_rms = max(noise_el, n_el), max(noise_rms, n_rms)
        over += int(el >= EL or rms >= RMS)
        same = bool(np.array_equal(model(x), y))
        for grid in (1, 3):
            model.set_option("fused_grid", grid)
            same = same and bool(np.array_equal(model(x), y))
        for frames, grid in ((8, 0), (8, 2), (4, 0)):      # both forms of the kernel (these shapes run the four-frame one by default)
            model.set_option("bf16_frames", frames)
            model.set_option("fused_grid", grid)
            same = same and bool(np.array_equal(model(x), y))
        model.set_option("bf16_frames", 0)
        if not (np.isfinite(y).all() and el < max(2 * EL, 3 * n_el) and rms < max(2 * RMS, 3 * n_rms) and same):
            failed.append({"weights_seed": wseed, "shape": [n, t], "scale": scale, "element": el, "rms": rms, "emulation_noise": [n_el, n_rms],
                           "bit_identical": same})
        worst_el, worst_rms = max(worst_el, el), max(worst_rms, rms)
        shapes.append((n, t))
        if i % 10 == 9:
            print("[fuzz bf16] %s %d/%d worst %.2e / rms %.2e (%.0f s)" % (net, i + 1, cases, worst_el, worst_rms, time.time() - t_start),
                  file=sys.stderr, flush=True)
    model.close()
    out[net] = {"cases": cases, "worst_element_error_of_scale": worst_el, "worst_rms_error_of_scale": worst_rms, "bounds": [EL, RMS],
                "cases_over_the_fixed_bounds": over, "emulation_fp64_vs_fp32_accumulation_worst": [noise_el, noise_rms], "failed": failed, "frames_min_max": [min(a * b for a, b in shapes), max(a * b for a, b in shapes)]}
print(json.dumps(out))
