"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the forward pass.

PARITY UNPINNED: restates /root/reference/model_utils/module.py:11-34 (conv_bn_relu)
and model_utils/model.py:6-96 (the three nets) from the code plus TensorFlow 1.14's
documented defaults; TensorFlow itself cannot run in the build container and the
reference holds no golden vectors, so nothing here was checked against the real thing.
What IS checked: the published parameter counts (readme.md:65-67), analytic
known-answer cases, and agreement between three independent restatements
(this file, oracle/rced_oracle.c, oracle/torch_ref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Layout everywhere: NHWC = [N, T(time), F(freq=129), C]; kernels HWIO = [kh, kw, cin, cout]
(module.py:15,27; data_loader.py:206-208).
"""

import numpy as np

from . import layers as L


def same_pad(k):
    """TF 'SAME', stride 1: total k-1, floor half before, rest after (k=8 -> 3 before, 4 after)."""
    total = k - 1
    return total // 2, total - total // 2


def conv2d_same(x, kernel, bias, dtype=np.float64):
    """tf.layers.conv2d(x, cout, (kh,kw), (1,1), 'SAME') with use_bias=True (module.py:27).

    Loop nest over taps; each tap is a [pixels, cin] x [cin, cout] product.
    """
    x = np.asarray(x, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    n, t, f, cin = x.shape
    kh, kw, cin2, cout = kernel.shape
    assert cin == cin2, (cin, cin2)
    pt, pb = same_pad(kh)
    pl, pr = same_pad(kw)
    xp = np.zeros((n, t + kh - 1, f + kw - 1, cin), dtype=dtype)
    xp[:, pt:pt + t, pl:pl + f, :] = x
    y = np.zeros((n, t, f, cout), dtype=dtype)
    for i in range(kh):
        for j in range(kw):
            y += xp[:, i:i + t, j:j + f, :] @ kernel[i, j]
    y += np.asarray(bias, dtype=dtype)
    return y


def batch_norm_inference(y, gamma, beta, mean, var, eps=L.BN_EPS, dtype=np.float64):
    """tf.layers.batch_normalization(training=False): (y-mean)*gamma/sqrt(var+eps)+beta (module.py:29)."""
    g, b, m, v = (np.asarray(a, dtype=dtype) for a in (gamma, beta, mean, var))
    return (y - m) * (g / np.sqrt(v + dtype(eps))) + b


def conv_bn_relu(x, kernel, bias, bn=None, skip_input=None, use_act=True, dtype=np.float64):
    """module.py:11-34: conv -> BN (if use_norm) -> + skip_input -> ReLU (if use_act)."""
    y = conv2d_same(x, kernel, bias, dtype)
    if bn is not None:
        y = batch_norm_inference(y, *bn, dtype=dtype)
    if skip_input is not None:
        y = y + np.asarray(skip_input, dtype=dtype)
    if use_act:
        y = np.maximum(y, 0)
    return y


def forward(net_work, weights, x, dtype=np.float64, return_all=False):
    """model(x) of model.py:26-29 / 58-61 / 93-96 with is_training=False.

    weights: dict keyed by TF variable name ("{scope}/kernel", ...; see layers.variable_shapes).
    x: [N, T, 129, 1].  Returns [N, T, 129, 1] (and every intermediate tensor if return_all).
    """
    layers = L.layers_for(net_work)
    tensors = [np.asarray(x, dtype=dtype)]
    for l in layers:
        bn = None
        if l.use_norm:
            p = l.scope + "/batch_norm/"
            bn = (weights[p + "gamma"], weights[p + "beta"], weights[p + "moving_mean"], weights[p + "moving_variance"])
        y = conv_bn_relu(tensors[l.src], weights[l.scope + "/kernel"], weights[l.scope + "/bias"], bn,
                         tensors[l.skip_pre] if l.skip_pre >= 0 else None, l.use_act, dtype)
        if l.skip_post >= 0:  # V3 simple_RCED: encode_3 + skip_input AFTER the ReLU (model.py:75-76)
            y = y + tensors[l.skip_post]
        tensors.append(y)
    return tensors if return_all else tensors[-1]


def bf16_round(a):
    """Round float32 values to bfloat16 (round to nearest, ties to even), returned as float32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32).reshape(np.shape(a))


def forward_bf16(net_work, weights, x, final_bf16=True, accumulate=np.float64):
    """Emulation of the bf16 mode of the R-CED kernels (csrc/kernels_frame16.h; BASELINE config 2).

    Not a statement about the reference (which is fp32): it restates what the bf16 KERNEL computes, so that the
    kernel can be tested.  The input is cast to bf16 ("C2 ... (cast bf16)"); every layer: BatchNorm-folded
    kernel rounded to bf16, bf16 activations in, fp32 (here fp64) accumulation, fp32 shift, (+ skip), ReLU, result
    rounded to bf16.  (Until round 6 the first layer ran on the fp32 input with its fp32 kernel.)  Last layer (1x129):
    kernel rounded to bf16 (frame16::run_final; final_bf16=False keeps it fp32, as an earlier form of the library could)
    on the bf16 activations, fp32 output.

    `accumulate`: the dtype the convolution sums run in.  A sum that lands within its own rounding error of a bf16 midpoint rounds
    either way, a layer's output then differs by one bf16 step (2^-8 of the value) and fifteen layers pass that on: two runs of THIS
    emulation that differ in nothing but the accumulation dtype (fp64 / fp32) are 3e-3 of the scale apart on typical inputs and
    1-2e-2 on some (small input scales, where the shifts dominate) -- the yardstick tests/tools/fuzz_bf16.py holds the kernel to.
    """
    layers = L.layers_for(net_work)
    tensors = [bf16_round(np.asarray(x, dtype=np.float32)).astype(np.float64)]
    for i, l in enumerate(layers):
        k = np.asarray(weights[l.scope + "/kernel"], np.float64)
        shift = np.asarray(weights[l.scope + "/bias"], np.float64)
        if l.use_norm:
            p = l.scope + "/batch_norm/"
            g, b, m, v = (np.asarray(weights[p + n], np.float64) for n in ("gamma", "beta", "moving_mean", "moving_variance"))
            s = g / np.sqrt(v + L.BN_EPS)
            k = k * s
            shift = (shift - m) * s + b
        k = k.astype(np.float32)                 # the folded kernel as the library holds it
        last = i == len(layers) - 1
        if not (last and not final_bf16):
            k = bf16_round(k)
        y = conv2d_same(tensors[l.src].astype(accumulate), k.astype(accumulate), shift.astype(np.float32).astype(accumulate),
                        accumulate).astype(np.float64)
        if l.skip_pre >= 0:
            y = y + tensors[l.skip_pre]
        if l.use_act:
            y = np.maximum(y, 0)
        if l.skip_post >= 0:
            y = y + tensors[l.skip_post]
        if not last:
            y = bf16_round(y.astype(np.float32)).astype(np.float64)
        tensors.append(y)
    return tensors[-1].astype(np.float32)


def make_weights(net_work, seed=42, trivial_bn=False):
    """Synthetic weights of SURVEY 8(d2): glorot-uniform kernels, non-trivial bias and BN stats.

    (TF's own init would be zero bias and identity BN; non-trivial values make folding bugs show.)
    """
    rng = np.random.default_rng(seed)
    layers = L.layers_for(net_work)
    w = {}
    for i, l in enumerate(layers):
        cin = L.cin_of(layers, i)
        fan_in, fan_out = l.kh * l.kw * cin, l.kh * l.kw * l.cout
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        w[l.scope + "/kernel"] = rng.uniform(-lim, lim, (l.kh, l.kw, cin, l.cout)).astype(np.float32)
        w[l.scope + "/bias"] = rng.uniform(-0.1, 0.1, l.cout).astype(np.float32)
        if l.use_norm:
            p = l.scope + "/batch_norm/"
            if trivial_bn:
                w[p + "gamma"] = np.ones(l.cout, np.float32)
                w[p + "beta"] = np.zeros(l.cout, np.float32)
                w[p + "moving_mean"] = np.zeros(l.cout, np.float32)
                w[p + "moving_variance"] = np.ones(l.cout, np.float32)
            else:
                w[p + "gamma"] = rng.uniform(0.5, 1.5, l.cout).astype(np.float32)
                w[p + "beta"] = rng.uniform(-0.1, 0.1, l.cout).astype(np.float32)
                w[p + "moving_mean"] = rng.normal(0, 0.1, l.cout).astype(np.float32)
                w[p + "moving_variance"] = rng.uniform(0.5, 1.5, l.cout).astype(np.float32)
    return w


def make_input(n, t, seed=1234, f=L.FEATURE_DIM):
    """|N(0,1)| magnitudes, SURVEY 8(d2)."""
    rng = np.random.default_rng(seed)
    return np.abs(rng.standard_normal((n, t, f, 1))).astype(np.float32)
