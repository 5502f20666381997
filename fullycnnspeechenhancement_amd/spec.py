"""Topology of the three networks, as the C ABI reports it (single source: csrc/rced_spec.h).

Mirrors what a reader of the reference gets from model_utils/model.py:6-96; the selection rule
`net_work in {"FullyCNNV2", "FullyCNNV3", else V1}` is infer.py:45-51 / tester.py:76-82.
"""

import ctypes
from collections import namedtuple

from . import _lib

FEATURE_DIM = 129
V1, V2, V3 = 1, 2, 3

LayerDesc = namedtuple("LayerDesc", "scope cout kh kw use_norm use_act src skip_pre skip_post cin")


def variant_of(net_work):
    """cfg [model] net_work -> variant id (anything that is not V2/V3 falls back to V1, as the reference does)."""
    if net_work in (V1, V2, V3):
        return net_work
    if net_work == "FullyCNNV2":
        return V2
    if net_work == "FullyCNNV3":
        return V3
    return V1


def layers(variant):
    lib = _lib.load()
    n = lib.rced_num_layers(variant)
    if n < 0:
        raise ValueError("unknown variant %r" % (variant,))
    out = []
    buf = (ctypes.c_int * 9)()
    for i in range(n):
        _lib.check(lib.rced_layer_desc(variant, i, buf))
        out.append(LayerDesc(lib.rced_layer_scope(variant, i).decode(), *list(buf)))
    return out


def variable_shapes(variant):
    """[(TF variable name, shape)] in blob order (module.py:27,29 naming)."""
    out = []
    for l in layers(variant):
        out.append((l.scope + "/kernel", (l.kh, l.kw, l.cin, l.cout)))
        out.append((l.scope + "/bias", (l.cout,)))
        if l.use_norm:
            for v in ("gamma", "beta", "moving_mean", "moving_variance"):
                out.append((l.scope + "/batch_norm/" + v, (l.cout,)))
    return out


def num_weights(variant):
    return int(_lib.load().rced_num_weights(variant))


def num_trainable(variant):
    """What BaseTester.param_count prints (tester.py:41-47): 32765 / 32192 / 32653."""
    return int(_lib.load().rced_num_trainable(variant))


def flops_per_frame(variant):
    """Nominal dense FLOPs per 129-bin frame (2 * MAC, zero-padded taps included): SURVEY 8(d3)."""
    return 2 * FEATURE_DIM * sum(l.kh * l.kw * l.cin * l.cout for l in layers(variant))
