"""ctypes binding of librced_hip.so (C ABI: include/rced.h).

The library is the product; there is NO CPU fallback.  If the .so is missing this module raises
at load() time -- build it with `python __graft_entry__.py` (or `make -C fullycnnspeechenhancement_amd/csrc`).
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCED_LIB: A/B experiments only (an alternative build of the same ABI); the product is librced_hip.so
SO_PATH = os.environ.get("RCED_LIB") or os.path.join(_HERE, "librced_hip.so")

RCED_OK, RCED_ERR_ARG, RCED_ERR_HIP, RCED_ERR_ALLOC, RCED_ERR_STATE = 0, 1, 2, 3, 4
PATH_AUTO, PATH_LAYERWISE, PATH_FUSED = 0, 1, 2
K_GENERIC, K_FUSED, K_FINAL = 0, 1, 2

# every symbol include/rced.h declares: (restype, argtypes)
_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
SYMBOLS = {
    "rced_num_layers": (ctypes.c_int, [ctypes.c_int]),
    "rced_num_weights": (ctypes.c_size_t, [ctypes.c_int]),
    "rced_num_trainable": (ctypes.c_size_t, [ctypes.c_int]),
    "rced_layer_desc": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _c_int_p]),
    "rced_layer_scope": (ctypes.c_char_p, [ctypes.c_int, ctypes.c_int]),
    "rced_create": (ctypes.c_int, [ctypes.c_int, _c_float_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(_vp)]),
    "rced_destroy": (None, [_vp]),
    "rced_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "rced_check": (ctypes.c_int, [_vp]),
    "rced_forward_host": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int]),
    "rced_reserve": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "rced_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    "rced_get_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _c_int_p]),
    "rced_conv_bn_relu": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [ctypes.c_int] * 9 + [_vp]),
    "rced_conv_bn_relu_train": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [ctypes.c_int] * 8 + [_vp, ctypes.c_int, _vp]),
    "rced_stft_num_frames": (ctypes.c_int, [ctypes.c_int]),
    "rced_stft": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp]),
    "rced_istft": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp]),
    "rced_stft_ex": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, ctypes.c_int]),
    "rced_istft_ex": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, ctypes.c_int]),
    "rced_train_create": (ctypes.c_int, [ctypes.c_int, _c_float_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(_vp)]),
    "rced_train_destroy": (None, [_vp]),
    "rced_train_step": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.POINTER(ctypes.c_double), _vp]),
    "rced_train_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "rced_train_global_step": (ctypes.c_longlong, [_vp]),
    "rced_train_get_variables": (ctypes.c_int, [_vp, _c_float_p, ctypes.c_size_t]),
    "rced_train_get_gradients": (ctypes.c_int, [_vp, _c_float_p, ctypes.c_size_t]),
    "rced_train_get_state": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_longlong)]),
    "rced_train_set_state": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, ctypes.c_size_t, ctypes.c_longlong]),
    "rced_last_kernel_ms": (ctypes.c_float, [_vp]),
    "rced_profile_query": (ctypes.c_int, [_vp, ctypes.c_int, _c_float_p, _c_int_p]),
    "rced_last_error": (ctypes.c_char_p, []),
    "rced_version": (ctypes.c_char_p, []),
}

_lib = None


class RcedError(RuntimeError):
    """Raised for every non-zero status from the C ABI (mirrors TF raising from sess.run)."""

    def __init__(self, code, msg):
        super().__init__("rced error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load librced_hip.so.  Import torch first when both live in one process, so that the
    one HIP runtime torch ships (same soname, libamdhip64.so.7) is the one we bind to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "%s not found: the HIP extension is not built (run `python __graft_entry__.py`). "
            "There is no CPU fallback for the R-CED forward pass." % SO_PATH)
    try:
        import torch  # noqa: F401  (binds torch's bundled libamdhip64 first)
    except Exception:  # pragma: no cover - torch-less use of the C ABI is fine
        pass
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != RCED_OK:
        raise RcedError(rc, load().rced_last_error().decode())


def version():
    return load().rced_version().decode()
