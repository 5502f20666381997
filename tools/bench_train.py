#!/usr/bin/env python3
"""BASELINE config 5: CR-CED V3 training step (fwd + bwd + Adam), batch 256 x 512 frames, one MI355X.
Prints ms per step (wall, synchronised) and the first losses.  Correctness-first kernels: see DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fullycnnspeechenhancement_amd import FullyCNNTrainer, spec, weights as _weights

NET = os.environ.get("NET", "FullyCNNV3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = _weights.synthetic_weights(spec.variant_of(NET), seed=42)
tr = FullyCNNTrainer(NET, batch_size=B, lr=1e-3, warmup_steps=4000.0, weights=w)
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn((B, T, 129, 1), generator=g, device="cuda").abs_()
y = 0.5 * torch.randn((B, T, 129, 1), generator=g, device="cuda").abs_()
losses, times = [], []
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l, _, s = tr.fit_step(x, y)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0); losses.append(l)
flop = 3 * 8207496 * B * T          # ~3x forward
print(json.dumps({"config": "%s train step, batch %d x %d" % (NET, B, T), "ms_per_step": 1e3 * min(times[1:]),
                  "first_step_ms": 1e3 * times[0], "losses": losses, "approx_tflops": flop / min(times[1:]) / 1e12,
                  "mem_GB": torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9}))
