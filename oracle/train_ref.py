"""ORACLE (test infrastructure, NOT product code): one training step of the reference trainer,
restated on PyTorch-CPU autograd.

PARITY UNPINNED: restates model_utils/trainer.py:143-192 (loss, optimizer, train_step) and :68-76 (Noam
learning rate) over the train-mode graph (is_training=True: BatchNorm normalises with the batch mean and
the biased batch variance and moves its statistics with momentum 0.99) from the reference source plus
TensorFlow-1.14 defaults; TensorFlow cannot run here.  TF specifics hard-coded:
  * loss = sum((target - pred)^2) / batch_size   with the CONFIGURED batch size (trainer.py:146-147,153)
  * tf.train.AdamOptimizer defaults beta1 0.9, beta2 0.999, eps 1e-8 in TF's form
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  theta -= lr_t * m / (sqrt(v) + eps)
  * moving_variance is updated with the UNBIASED batch variance (TF fused batch-norm), moving_mean
    with the batch mean, both `moving = 0.99 * moving + 0.01 * batch`  (cannot be confirmed here).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

import numpy as np
import torch
import torch.nn.functional as Fn

from . import layers as L

BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8


def noam_lr(init_lr, global_step, warmup_steps):
    """trainer.py:68-76."""
    step = global_step + 1
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


def trainable_names(net_work):
    return [n for n, _ in L.variable_shapes(L.layers_for(net_work)) if "moving_" not in n]


class _TapsConv(torch.autograd.Function):
    """y[n,t,f,:] = sum_{i,j} xp[n,t+i,f+j,:] @ k[i,j] on NHWC tensors (xp = the SAME-padded input), tap by tap with plain
    matmuls.  A custom Function because autograd through the plain loop keeps one contiguous copy of the shifted input PER
    TAP for the backward (129 copies for the 1x129 output layer: 139 GB at BASELINE config 5's size); this one keeps x
    and k only and walks the taps again in backward."""

    @staticmethod
    def forward(ctx, x, k, pads):
        pt, pb, pl, pr = pads
        kh, kw = k.shape[0], k.shape[1]
        n, t, f, _ = x.shape
        xp = Fn.pad(x, (0, 0, pl, pr, pt, pb))
        y = None
        for i in range(kh):
            for j in range(kw):
                term = xp[:, i:i + t, j:j + f, :] @ k[i, j]
                y = term if y is None else y.add_(term)
        ctx.save_for_backward(x, k)
        ctx.pads = pads
        return y

    @staticmethod
    def backward(ctx, g):
        x, k = ctx.saved_tensors
        pt, pb, pl, pr = ctx.pads
        kh, kw, cin, cout = k.shape
        n, t, f, _ = x.shape
        xp = Fn.pad(x, (0, 0, pl, pr, pt, pb))
        g2 = g.reshape(-1, cout)
        dk = torch.empty_like(k)
        dxp = torch.zeros_like(xp) if ctx.needs_input_grad[0] else None
        for i in range(kh):
            for j in range(kw):
                if ctx.needs_input_grad[1]:
                    dk[i, j] = xp[:, i:i + t, j:j + f, :].reshape(-1, cin).t() @ g2
                if dxp is not None:
                    dxp[:, i:i + t, j:j + f, :] += g @ k[i, j].t()
        dx = dxp[:, pt:pt + t, pl:pl + f, :] if dxp is not None else None
        return dx, (dk if ctx.needs_input_grad[1] else None), None


class TrainRef:
    def __init__(self, net_work, weights, batch_size, dtype=torch.float64, device="cpu", conv="conv2d"):
        """device / conv: the default is F.conv2d on the CPU.  conv="taps" computes every convolution as a sum over
        kernel taps of [pixels, cin] x [cin, cout] matmuls on NHWC tensors -- plain torch matmuls, which also run in
        float64 on a GPU (tests use it to restate BASELINE config 5 at full size, where batch statistics need the
        whole 256 x 512 batch)."""
        self.layers = L.layers_for(net_work)
        self.net_work = net_work
        self.batch_size = batch_size
        self.dtype = dtype
        self.device = torch.device(device)
        self.conv = conv
        self.vars = {k: torch.tensor(np.asarray(v), dtype=dtype, device=self.device) for k, v in weights.items()}
        for k in trainable_names(net_work):
            self.vars[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.vars[k]) for k in trainable_names(net_work)}
        self.v = {k: torch.zeros_like(self.vars[k]) for k in trainable_names(net_work)}
        self.global_step = 0

    def _forward_train_taps(self, x, keep_preact=False, leaf_after=None):
        """The same graph on NHWC tensors with tap-wise matmuls (module.py:27 `SAME`: (k-1)//2 before, the rest after).
        leaf_after = a layer scope: everything up to and including that layer runs without autograd, its output becomes a
        leaf that requires grad (self.mid = (leaf, normalised pre-activation zhat, BatchNorm output u) of that layer), and
        the rest of the net is recorded -- d loss / d leaf then gives that layer's d beta = sum g [u > 0] and
        d gamma = sum g [u > 0] zhat without holding the whole net's graph (used at BASELINE config 5's full size)."""
        tens = [torch.as_tensor(x).to(device=self.device, dtype=self.dtype)]
        stats, pre = [], []
        recording = leaf_after is None
        self.mid = None
        for l in self.layers:
            with torch.set_grad_enabled(recording and torch.is_grad_enabled()):
                k = self.vars[l.scope + "/kernel"]                                    # HWIO
                pt, pb = (l.kh - 1) // 2, (l.kh - 1) - (l.kh - 1) // 2
                pl, pr = (l.kw - 1) // 2, (l.kw - 1) - (l.kw - 1) // 2
                src = tens[l.src]
                n, t, f, _ = src.shape
                y = _TapsConv.apply(src, k, (pt, pb, pl, pr))
                y = y + self.vars[l.scope + "/bias"]
                zhat = None
                if l.use_norm:
                    p = l.scope + "/batch_norm/"
                    mean = y.mean(dim=(0, 1, 2))
                    var = ((y - mean) ** 2).mean(dim=(0, 1, 2))
                    zhat = (y - mean) / torch.sqrt(var + L.BN_EPS)
                    y = zhat * self.vars[p + "gamma"] + self.vars[p + "beta"]
                    stats.append((l.scope, mean.detach(), var.detach(), y.numel() // y.shape[-1]))
                if l.skip_pre >= 0:
                    y = y + tens[l.skip_pre]
                u = y
                if l.use_act:
                    if keep_preact:
                        pre.append(float(y.detach().abs().min()))
                    y = torch.relu(y)
                if l.skip_post >= 0:
                    y = y + tens[l.skip_post]
            if leaf_after is not None and l.scope == leaf_after:
                y = y.detach().requires_grad_(True)
                self.mid = (y, zhat.detach(), u.detach())
                recording = True
            tens.append(y)
        self.last_hidden = tens[-2]
        return (tens[-1], stats, pre) if keep_preact else (tens[-1], stats)

    def mid_layer_bn_grads(self, x, target, scope):
        """(d beta, d gamma) of the plain conv+BN+ReLU layer `scope` (no skip in or out), from the gradient that reaches
        its output through the rest of the net: d beta = sum g [u > 0], d gamma = sum g [u > 0] zhat
        (module.py:28-33; tf.layers.batch_normalization's own backward).  Also returns the loss."""
        pred, _ = self._forward_train_taps(x, leaf_after=scope)
        loss = ((torch.as_tensor(target).to(device=pred.device, dtype=self.dtype) - pred) ** 2).sum() / self.batch_size
        leaf, zhat, u = self.mid
        (g,) = torch.autograd.grad(loss, leaf)
        du = g * (u > 0)
        return du.sum(dim=(0, 1, 2)), (du * zhat).sum(dim=(0, 1, 2)), float(loss.detach())

    def min_abs_preactivation(self, x):
        """Smallest |value entering a ReLU| over the whole net for this input: a ReLU whose input is within fp32
        rounding of zero may take the other branch in an fp32 implementation, which changes gradients by whole
        terms.  Tests screen their inputs with it (reject-and-redraw) so that tight gradient bounds are meaningful."""
        with torch.no_grad():
            ref = TrainRef(self.net_work, {k: v.detach().cpu().numpy() for k, v in self.vars.items()}, self.batch_size,
                           self.dtype, self.device, "taps")
            return min(ref._forward_train_taps(x, keep_preact=True)[2])

    def forward_train(self, x):
        """model(x) with is_training=True; returns (pred, list of (scope, batch_mean, batch_var_biased, count))."""
        if self.conv == "taps":
            return self._forward_train_taps(x)
        tens = [torch.as_tensor(x).to(self.dtype).permute(0, 3, 1, 2)]
        stats = []
        for l in self.layers:
            k = self.vars[l.scope + "/kernel"].permute(3, 2, 0, 1).contiguous()   # HWIO -> OIHW
            pt, pb = (l.kh - 1) // 2, (l.kh - 1) - (l.kh - 1) // 2
            pl, pr = (l.kw - 1) // 2, (l.kw - 1) - (l.kw - 1) // 2
            y = Fn.conv2d(Fn.pad(tens[l.src], (pl, pr, pt, pb)), k, self.vars[l.scope + "/bias"])
            if l.use_norm:
                p = l.scope + "/batch_norm/"
                mean = y.mean(dim=(0, 2, 3), keepdim=True)
                var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
                y = (y - mean) / torch.sqrt(var + L.BN_EPS) * self.vars[p + "gamma"].view(1, -1, 1, 1) + \
                    self.vars[p + "beta"].view(1, -1, 1, 1)
                stats.append((l.scope, mean.detach().flatten(), var.detach().flatten(), y.numel() // y.shape[1]))
            if l.skip_pre >= 0:
                y = y + tens[l.skip_pre]
            if l.use_act:
                y = torch.relu(y)
            if l.skip_post >= 0:
                y = y + tens[l.skip_post]
            tens.append(y)
        return tens[-1].permute(0, 2, 3, 1), stats

    def loss_and_grads(self, x, target):
        for k in self.m:
            self.vars[k].grad = None
        pred, stats = self.forward_train(x)
        loss = ((torch.as_tensor(target).to(device=pred.device, dtype=self.dtype) - pred) ** 2).sum() / self.batch_size
        loss.backward()
        return loss.item(), {k: self.vars[k].grad.clone() for k in self.m}, stats

    def train_step(self, x, target, lr):
        """trainer.py:181-192 (+ UPDATE_OPS): returns (batch_loss, global_step after the step)."""
        loss, grads, stats = self.loss_and_grads(x, target)
        self.global_step += 1
        t = self.global_step
        lr_t = lr * np.sqrt(1 - BETA2 ** t) / (1 - BETA1 ** t)
        with torch.no_grad():
            for k, g in grads.items():
                self.m[k].mul_(BETA1).add_(g, alpha=1 - BETA1)
                self.v[k].mul_(BETA2).addcmul_(g, g, value=1 - BETA2)
                self.vars[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + ADAM_EPS))
            for scope, mean, var, n in stats:
                p = scope + "/batch_norm/"
                self.vars[p + "moving_mean"].mul_(L.BN_MOMENTUM).add_(mean, alpha=1 - L.BN_MOMENTUM)
                self.vars[p + "moving_variance"].mul_(L.BN_MOMENTUM).add_(var * n / max(n - 1, 1), alpha=1 - L.BN_MOMENTUM)
        return loss, self.global_step

    def weights(self):
        return {k: v.detach().numpy().copy() for k, v in self.vars.items()}
