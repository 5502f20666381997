"""ORACLE (test infrastructure): ctypes binding of oracle/librced_oracle.so (rced_oracle.c).

PARITY UNPINNED -- see rced_oracle.c.  Only tests/, smoke() and bench.py's cpu_baseline import this.
"""

import ctypes
import os
import subprocess

import numpy as np

from . import layers as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librced_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "rced_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        for name, real in (("oracle_forward_f64", ctypes.c_double), ("oracle_forward_f32", ctypes.c_float)):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                           ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                           ctypes.POINTER(real)]
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
    return _lib


def descriptor(layers):
    d = []
    for i, l in enumerate(layers):
        d += [l.cout, l.kh, l.kw, int(l.use_norm), int(l.use_act), l.src, l.skip_pre, l.skip_post, L.cin_of(layers, i)]
    return np.asarray(d, dtype=np.int32)


def pack_blob(layers, weights):
    parts = [np.asarray(weights[name], dtype=np.float32).reshape(-1) for name, _ in L.variable_shapes(layers)]
    return np.ascontiguousarray(np.concatenate(parts))


def forward(net_work, weights, x, dtype=np.float64):
    """model(x), is_training=False.  x [N,T,F,1] float32 -> [N,T,F,1] of `dtype` (float64 checker / float32 port)."""
    layers = L.layers_for(net_work)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, t, f, c = x.shape
    assert c == 1
    desc = descriptor(layers)
    blob = pack_blob(layers, weights)
    y = np.empty((n, t, f, 1), dtype=dtype)
    fn, ct = (lib().oracle_forward_f64, ctypes.c_double) if dtype == np.float64 else (lib().oracle_forward_f32, ctypes.c_float)
    rc = fn(desc.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(layers),
            blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            n, t, f, y.ctypes.data_as(ctypes.POINTER(ct)))
    if rc != 0:
        raise RuntimeError("oracle_forward failed: %d" % rc)
    return y


def num_threads():
    return lib().oracle_num_threads()
