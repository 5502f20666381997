"""Weight container keyed by the reference's TF variable names.

The reference keeps weights inside the tf.Session and fills them with
`Saver.restore(sess, checkpoint)` (model_utils/tester.py:36-39) or
`tf.global_variables_initializer()` (model_utils/trainer.py:60-65).  Here they are a dict
{"{scope}/kernel": [kh,kw,cin,cout], "{scope}/bias": [cout],
 "{scope}/batch_norm/{gamma,beta,moving_mean,moving_variance}": [cout]}  (module.py:27,29)
that an offline checkpoint dump (.npz with those keys) drops into.
"""

import numpy as np

from . import spec


def initial_weights(variant, seed=None):
    """TF-1.14 initial values: glorot-uniform kernels, zero bias, BN gamma=1 beta=0 mean=0 var=1."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in spec.variable_shapes(variant):
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            kh, kw, cin, cout = shape
            lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf in ("gamma", "moving_variance"):
            w[name] = np.ones(shape, np.float32)
        else:
            w[name] = np.zeros(shape, np.float32)
    return w


def synthetic_weights(variant, seed=42):
    """Random weights for benchmarks and smoke runs: glorot-uniform kernels, bias U(-0.1, 0.1),
    non-trivial BatchNorm (gamma U(0.5, 1.5), beta U(-0.1, 0.1), moving_mean N(0, 0.1), moving_variance
    U(0.5, 1.5)) so that a folding bug changes the result.  Draw order = variable order, one generator."""
    rng = np.random.default_rng(seed)
    w = {}
    for l in spec.layers(variant):
        lim = np.sqrt(6.0 / (l.kh * l.kw * l.cin + l.kh * l.kw * l.cout))
        w[l.scope + "/kernel"] = rng.uniform(-lim, lim, (l.kh, l.kw, l.cin, l.cout)).astype(np.float32)
        w[l.scope + "/bias"] = rng.uniform(-0.1, 0.1, l.cout).astype(np.float32)
        if l.use_norm:
            p = l.scope + "/batch_norm/"
            w[p + "gamma"] = rng.uniform(0.5, 1.5, l.cout).astype(np.float32)
            w[p + "beta"] = rng.uniform(-0.1, 0.1, l.cout).astype(np.float32)
            w[p + "moving_mean"] = rng.normal(0, 0.1, l.cout).astype(np.float32)
            w[p + "moving_variance"] = rng.uniform(0.5, 1.5, l.cout).astype(np.float32)
    return w


def validate(variant, weights):
    """Shape / dtype / finiteness check with reference-style messages (TF raises on restore mismatch)."""
    for name, shape in spec.variable_shapes(variant):
        if name not in weights:
            raise KeyError("weight %r missing (variant %d)" % (name, variant))
        a = np.asarray(weights[name])
        if tuple(a.shape) != tuple(shape):
            raise ValueError("weight %r has shape %s, expected %s" % (name, a.shape, shape))
        if not np.all(np.isfinite(a)):
            raise ValueError("weight %r has non-finite values" % name)


def pack_blob(variant, weights):
    """Flatten to the float32 blob rced_create takes (order = spec.variable_shapes)."""
    validate(variant, weights)
    parts = [np.asarray(weights[name], dtype=np.float32).reshape(-1) for name, _ in spec.variable_shapes(variant)]
    blob = np.ascontiguousarray(np.concatenate(parts))
    assert blob.size == spec.num_weights(variant)
    return blob


def load(path, variant):
    """What `Saver.restore(sess, checkpoint_file)` does for the reference (tester.py:36-39): accepts a
    TF V2 checkpoint prefix (or its .index), a frozen .pb (freeze.py) or the .npz interchange file."""
    from . import tf_checkpoint
    w = tf_checkpoint.load_reference_weights(path, variant)
    validate(variant, w)
    return w


def load_npz(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def save_npz(path, weights):
    np.savez(path, **{k: np.asarray(v) for k, v in weights.items()})
