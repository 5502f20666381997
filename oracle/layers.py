"""ORACLE (test infrastructure, NOT product code) -- layer tables of the three nets.

PARITY UNPINNED: the reference hot path is TensorFlow 1.14 graph code
(model_utils/module.py:7-8) and TensorFlow cannot be imported in the build
container; the reference ships no tests or golden vectors.  These tables are a
hand transcription of /root/reference/model_utils/model.py, pinned only by the
three parameter counts the reference publishes (readme.md:65-67).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

A layer is (scope, cout, kh, kw, use_norm, use_act, src, skip_pre, skip_post):
  src       tensor id the conv reads (0 = network input, i+1 = output of layer i)
  skip_pre  tensor id added after BN and BEFORE ReLU  (module.py:30-31), or -1
  skip_post tensor id added AFTER ReLU (model.py:75-76, V3 block skip), or -1
"""

from collections import namedtuple

Layer = namedtuple("Layer", "scope cout kh kw use_norm use_act src skip_pre skip_post")

FEATURE_DIM = 129  # cfg [data] feature_dim; nfft 256 -> 129 bins (data_loader.py:59)
BN_EPS = 1e-3      # tf.layers.batch_normalization default epsilon (module.py:29 passes none)
BN_MOMENTUM = 0.99  # tf.layers.batch_normalization default momentum


def v1_layers():
    """R-CED 10 layers: FullyCNNSEModel, model_utils/model.py:6-29."""
    L = []
    # encode(): model.py:10-16.  Note the 5th encoder scope is "encode_8" (model.py:15).
    L.append(Layer("encode_1", 12, 8, 13, True, True, 0, -1, -1))
    L.append(Layer("encode_2", 16, 1, 11, True, True, 1, -1, -1))
    L.append(Layer("encode_3", 20, 1, 9, True, True, 2, -1, -1))
    L.append(Layer("encode_4", 24, 1, 7, True, True, 3, -1, -1))
    L.append(Layer("encode_8", 32, 1, 7, True, True, 4, -1, -1))
    # decode(): model.py:18-24, skip_input = encode_4..encode_1 (tensor ids 4..1)
    L.append(Layer("decode_1", 24, 1, 7, True, True, 5, 4, -1))
    L.append(Layer("decode_2", 20, 1, 9, True, True, 6, 3, -1))
    L.append(Layer("decode_3", 16, 1, 11, True, True, 7, 2, -1))
    L.append(Layer("decode_4", 12, 1, 13, True, True, 8, 1, -1))
    L.append(Layer("decode_5", 1, 1, 129, False, False, 9, -1, -1))
    return L


def v2_layers():
    """R-CED 16 layers: FullyCNNSEModelV2, model_utils/model.py:32-61."""
    enc = [(10, 8, 11), (12, 1, 7), (14, 1, 5), (15, 1, 5), (19, 1, 5), (21, 1, 5), (23, 1, 7), (25, 1, 11)]
    L = []
    for i, (c, kh, kw) in enumerate(enc):  # model.py:36-45
        L.append(Layer("encode_%d" % (i + 1), c, kh, kw, True, True, i, -1, -1))
    dec = [(23, 7, 7), (21, 5, 6), (19, 5, 5), (15, 5, 4), (14, 5, 3), (12, 7, 2), (10, 11, 1)]
    for j, (c, kw, skip) in enumerate(dec):  # model.py:47-55, skip = encode_7..encode_1
        L.append(Layer("decode_%d" % (j + 1), c, 1, kw, True, True, 8 + j, skip, -1))
    L.append(Layer("decode_8", 1, 1, 129, False, False, 15, -1, -1))  # model.py:55
    return L


def v3_layers():
    """CR-CED 16 layers: FullyCNNSEModelV3, model_utils/model.py:64-96."""
    L = []

    def block(name, first_kernel, src, skip_post):  # simple_RCED, model.py:68-78
        kh, kw = first_kernel
        L.append(Layer("%s_encode_1" % name, 18, kh, kw, True, True, src, -1, -1))
        L.append(Layer("%s_encode_2" % name, 30, 1, 5, True, True, len(L), -1, -1))
        L.append(Layer("%s_decode" % name, 8, 1, 9, True, True, len(L), -1, skip_post))
        return len(L)  # tensor id of the block output

    ce1 = block("CE1", (8, 9), 0, -1)       # model.py:81
    ce2 = block("CE2", (1, 9), ce1, -1)     # model.py:82
    ce3 = block("CE3", (1, 9), ce2, -1)     # model.py:83
    cd1 = block("CD1", (1, 9), ce3, ce2)    # model.py:87
    cd2 = block("CD2", (1, 9), cd1, ce1)    # model.py:88
    L.append(Layer("decode_final", 1, 1, 129, False, False, cd2, -1, -1))  # model.py:89-90
    return L


VARIANTS = {"FullyCNN": v1_layers, "FullyCNNV2": v2_layers, "FullyCNNV3": v3_layers}


def layers_for(net_work):
    """net_work selection rule of infer.py:45-51: V2, V3, anything else -> V1."""
    if net_work == "FullyCNNV2":
        return v2_layers()
    if net_work == "FullyCNNV3":
        return v3_layers()
    return v1_layers()


def cin_of(layers, i):
    src = layers[i].src
    return 1 if src == 0 else layers[src - 1].cout


def param_count(layers):
    """Trainable scalars as trainer.py:78-84 counts them: conv kernel+bias, BN gamma+beta."""
    n = 0
    for i, l in enumerate(layers):
        n += l.kh * l.kw * cin_of(layers, i) * l.cout + l.cout
        if l.use_norm:
            n += 2 * l.cout
    return n


def variable_shapes(layers):
    """(name, shape) in checkpoint order: TF variable names of module.py:27,29."""
    out = []
    for i, l in enumerate(layers):
        cin = cin_of(layers, i)
        out.append((l.scope + "/kernel", (l.kh, l.kw, cin, l.cout)))
        out.append((l.scope + "/bias", (l.cout,)))
        if l.use_norm:
            for v in ("gamma", "beta", "moving_mean", "moving_variance"):
                out.append((l.scope + "/batch_norm/" + v, (l.cout,)))
    return out
