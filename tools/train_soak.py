import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from fullycnnspeechenhancement_amd import FullyCNNTrainer, weights, spec
NET = os.environ.get("NET", "FullyCNNV3")   # FullyCNN | FullyCNNV2 | FullyCNNV3
w = weights.synthetic_weights({"FullyCNNV2": 2, "FullyCNNV3": 3}.get(NET, 1), seed=42)
tr = FullyCNNTrainer(NET, batch_size=256, lr=1e-3, warmup_steps=100.0, weights=w)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn((256, 512, 129, 1), generator=g, device="cuda").abs_()
y = 0.7 * x
losses = []
t0 = time.perf_counter()
for i in range(60):
    l, _, s = tr.fit_step(x, y)
    losses.append(l)
torch.cuda.synchronize()
v = tr.variables()
print(json.dumps({"net": NET, "steps": 60, "s_per_step": (time.perf_counter() - t0) / 60, "loss_first": losses[0], "loss_10": losses[10], "loss_last": losses[-1],
                  "finite": bool(all(np.isfinite(l) for l in losses) and all(np.isfinite(a).all() for a in v.values())),
                  "monotone_tail": bool(losses[-1] < losses[20] < losses[5])}))
