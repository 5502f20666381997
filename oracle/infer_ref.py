"""ORACLE (test infrastructure, NOT product code): the inference forward pass restated with plain torch matmuls, so that
it runs in float64 ON THE GPU BOX and can check EVERY output of a full-size forward (BASELINE config 3: 131,072 frames),
where the numpy / C restatements only afford samples.

PARITY UNPINNED (see oracle/layers.py): follows model_utils/module.py:11-34 (conv2d SAME + bias -> BatchNorm with the
moving statistics, eps 1e-3 -> + skip_input -> ReLU) and model_utils/model.py:6-96 (the three layer tables; V3's block
skip added AFTER the ReLU, model.py:75-76).  It is the inference twin of oracle/train_ref.py's conv="taps" path: every
convolution is a sum over kernel taps of [pixels, cin] x [cin, cout] matmuls on NHWC tensors; SAME padding puts
(k-1)//2 before and the rest after (module.py:27).  tests/test_oracle.py holds it to the numpy restatement (1e-12).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

import numpy as np
import torch
import torch.nn.functional as Fn

from . import layers as L


def _conv_taps(x, k):
    """x [n, t, f, cin], k [kh, kw, cin, cout] (HWIO) -> [n, t, f, cout], stride 1, SAME."""
    kh, kw = k.shape[0], k.shape[1]
    n, t, f, _ = x.shape
    pt, pb = (kh - 1) // 2, (kh - 1) - (kh - 1) // 2
    pl, pr = (kw - 1) // 2, (kw - 1) - (kw - 1) // 2
    xp = Fn.pad(x, (0, 0, pl, pr, pt, pb))
    y = None
    for i in range(kh):
        for j in range(kw):
            term = xp[:, i:i + t, j:j + f, :] @ k[i, j]
            y = term if y is None else y.add_(term)
    return y


@torch.no_grad()
def forward(net_work, weights, x, device="cpu", dtype=torch.float64, utterances_per_chunk=16):
    """model(x) with is_training=False.  x: ndarray or tensor [N, T, 129, 1]; returns a tensor [N, T, 129, 1] of `dtype` on
    `device`.  Utterances are independent in inference, so the batch is walked in chunks (memory: the widest net keeps
    114 skip channels alive)."""
    layers = L.layers_for(net_work)
    dev = torch.device(device)
    v = {k: torch.as_tensor(np.asarray(w)).to(device=dev, dtype=dtype) for k, w in weights.items()}
    x = torch.as_tensor(x)
    out = torch.empty(x.shape, dtype=dtype, device=dev)
    for n0 in range(0, x.shape[0], utterances_per_chunk):
        tens = [x[n0:n0 + utterances_per_chunk].to(device=dev, dtype=dtype)]
        for l in layers:
            y = _conv_taps(tens[l.src], v[l.scope + "/kernel"]) + v[l.scope + "/bias"]
            if l.use_norm:
                p = l.scope + "/batch_norm/"
                y = (y - v[p + "moving_mean"]) / torch.sqrt(v[p + "moving_variance"] + L.BN_EPS) * v[p + "gamma"] + v[p + "beta"]
            if l.skip_pre >= 0:
                y = y + tens[l.skip_pre]
            if l.use_act:
                y = torch.relu(y)
            if l.skip_post >= 0:
                y = y + tens[l.skip_post]
            tens.append(y)
        out[n0:n0 + utterances_per_chunk] = tens[-1]
    return out
