"""ORACLE (test infrastructure): PyTorch-CPU fp32 restatement of the forward pass.

PARITY UNPINNED -- this is NOT the TensorFlow reference (which cannot run here); it is a third
independent restatement of model_utils/module.py:11-34 + model_utils/model.py:6-96 on torch CPU ops,
used (a) to cross-check the numpy / C oracles and (b) as the multi-threaded CPU baseline of bench.py
(oneDNN convolutions are the closest stand-in for TF-1.14's Eigen/MKL CPU conv).
Only tests/, smoke() and bench.py's cpu_baseline leg may import this.
"""

import numpy as np
import torch
import torch.nn.functional as Fn

from . import layers as L


def _same_pad(k):
    return (k - 1) // 2, (k - 1) - (k - 1) // 2  # TF SAME: floor half before (k=8 -> 3, 4)


class TorchRef:
    def __init__(self, net_work, weights, dtype=torch.float32):
        self.layers = L.layers_for(net_work)
        self.dtype = dtype
        self.p = []
        for l in self.layers:
            k = torch.from_numpy(np.asarray(weights[l.scope + "/kernel"])).to(dtype)  # HWIO
            k = k.permute(3, 2, 0, 1).contiguous()                                     # -> OIHW
            b = torch.from_numpy(np.asarray(weights[l.scope + "/bias"])).to(dtype)
            bn = None
            if l.use_norm:
                q = l.scope + "/batch_norm/"
                bn = tuple(torch.from_numpy(np.asarray(weights[q + v])).to(dtype)
                           for v in ("gamma", "beta", "moving_mean", "moving_variance"))
            self.p.append((k, b, bn))

    @torch.no_grad()
    def __call__(self, x):
        """x: [N,T,F,1] (numpy or torch) -> torch [N,T,F,1]."""
        x = torch.as_tensor(x).to(self.dtype)
        tens = [x.permute(0, 3, 1, 2)]  # NHWC -> NCHW view: H=time, W=freq
        for l, (k, b, bn) in zip(self.layers, self.p):
            pt, pb = _same_pad(l.kh)
            pl, pr = _same_pad(l.kw)
            y = Fn.conv2d(Fn.pad(tens[l.src], (pl, pr, pt, pb)), k, b)
            if bn is not None:
                g, be, m, v = bn
                y = Fn.batch_norm(y, m, v, g, be, training=False, eps=L.BN_EPS)
            if l.skip_pre >= 0:
                y = y + tens[l.skip_pre]
            if l.use_act:
                y = torch.relu(y)
            if l.skip_post >= 0:
                y = y + tens[l.skip_post]
            tens.append(y)
        return tens[-1].permute(0, 2, 3, 1).contiguous()
