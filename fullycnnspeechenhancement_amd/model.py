"""Host-side mirror of the reference's model classes, over the C ABI (include/rced.h).

Reference surface kept (model_utils/model.py):
    FullyCNNSEModel(is_training)(x)     model.py:6-29   R-CED 10 layers
    FullyCNNSEModelV2(is_training)(x)   model.py:32-61  R-CED 16 layers
    FullyCNNSEModelV3(is_training)(x)   model.py:64-96  CR-CED 16 layers
x is [N, T, 129, 1] float32 NHWC and the result has the same shape.  In the reference `model(x)`
adds nodes to a TF graph and `sess.run` executes them (tester.py:85-90); here `model(x)` executes:
  * numpy in  -> numpy out   (host buffers, H2D/D2H inside; the reference's own calling convention)
  * torch.cuda tensor in -> torch.cuda tensor out (device-resident, zero copy, current stream)
Model(is_training=True)(x) is the graph trainer.py:165-172 builds: BatchNorm with batch statistics.
Weights are the TF variables by name (weights.py); `restore()` plays Saver.restore.
There is no CPU path: a missing HIP extension or GPU raises.
"""

import ctypes

import numpy as np

from . import _lib, spec, weights as _weights


class HostOutputPool(object):
    """Output arrays for the numpy-in / numpy-out boundary (`sess.run` returns a fresh ndarray per call, tester.py:85-90) without
    a fresh allocation per call: at config 3 a new 67 MB ndarray is ~5 ms of page faults on top of a 6.7 ms forward.

    `take(shape)` returns a NEW ndarray object every call -- a view of a backing buffer whose pages were touched when it was
    first handed out.  A backing buffer is handed out again only when nothing else refers to it any more: every view a caller
    still holds (or anything sliced from one) keeps a reference to it, which `sys.getrefcount` shows.  So the arrays behave
    like fresh ones -- a result the caller keeps is never written again -- and a caller that drops its results (the
    reference's loop: tester.py:104-113 consumes `pred_mag` before the next batch) gets the same warm pages every call.
    A caller that keeps everything gets new buffers up to `max_bytes` in all, then plain `numpy.empty`."""

    def __init__(self, max_bytes=1 << 30, per_shape=4):
        self.max_bytes, self.per_shape, self._bufs, self._bytes = int(max_bytes), int(per_shape), {}, 0

    def take(self, shape, dtype=np.float32):
        import sys
        key = (tuple(int(s) for s in shape), np.dtype(dtype).str)
        bufs = self._bufs.setdefault(key, [])
        for b in bufs:
            if sys.getrefcount(b) == 3:     # the list, the loop variable, getrefcount's own argument: no view is alive
                return b.view()
        nbytes = int(np.prod(key[0], dtype=np.int64)) * np.dtype(dtype).itemsize
        if len(bufs) >= self.per_shape or self._bytes + nbytes > self.max_bytes:
            if len(self._bufs) > 8:         # many shapes (variable-length utterances): forget the oldest shape's buffers
                old = next(iter(self._bufs))
                if old != key:
                    self._bytes -= sum(o.nbytes for o in self._bufs.pop(old))
            return np.empty(key[0], dtype)
        b = np.empty(key[0], dtype)
        bufs.append(b)
        self._bytes += nbytes
        return b.view()


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


class _RcedNet(object):
    variant = None
    _handle = None
    _train = None
    _host_pool = None

    def __init__(self, is_training, weights=None, device=0, seed=None, batch_size=1):
        """is_training=False: the inference graph of tester.py:69-83 / infer.py:36-52 (BatchNorm with the moving
        statistics; the fused forward kernels).  is_training=True: the graph trainer.py:165-172 builds --
        `model(x)` normalises every layer with the statistics of the batch it is given and updates nothing (only
        `pred` is fetched; the UPDATE_OPS and the optimizer belong to train_op).  It runs on the training-step
        library object (rced_train_forward); FullyCNNTrainer shares that handle for train_step.  batch_size is
        the configured size the trainer's loss divides by (trainer.py:146-147), unused by `model(x)` itself."""
        self.is_training = bool(is_training)
        self.device = int(device)
        self.batch_size = int(batch_size)
        self._handle = None
        self._train = None
        self._weights = None
        self._host_pool = HostOutputPool()   # numpy results: new ndarray objects over recycled, warm pages (None: numpy.empty per call)
        self._options = {}       # everything set through set_option / set_path: replayed by restore()
        self.restore(weights if weights is not None else _weights.initial_weights(self.variant, seed))

    # -- weights ---------------------------------------------------------------------------
    def restore(self, weights):
        """Saver.restore analogue (tester.py:36-39): dict keyed by TF variable names, or a .npz path.
        Options set earlier (path, bf16, host_chunks, fused_grid, ...) carry over to the new handle."""
        if isinstance(weights, str):
            weights = _weights.load_npz(weights)
        blob = _weights.pack_blob(self.variant, weights)
        lib = _lib.load()
        h = ctypes.c_void_p()
        fp = blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        if self.is_training:
            # A new library object: variables = `weights`, Adam slots and global_step start from zero -- what
            # Saver.restore of a checkpoint WITHOUT optimizer slots gives.  FullyCNNTrainer looks the handle up on every
            # call (it never caches the pointer), so trainer.model.restore(w) is safe; to carry optimizer state over use
            # FullyCNNTrainer.optimizer_state() / load_optimizer_state() or from_checkpoint().
            _lib.check(lib.rced_train_create(self.variant, fp, blob.size, self.batch_size, self.device, ctypes.byref(h)))
            self._release()
            self._train = h
        else:
            _lib.check(lib.rced_create(self.variant, fp, blob.size, self.device, ctypes.byref(h)))
            self._release()
            self._handle = h
            for key, value in self._options.items():
                if key != "profile":
                    _lib.check(lib.rced_set_option(self._handle, key.encode(), int(value)))
        self._weights = {k: np.array(v, dtype=np.float32) for k, v in weights.items()}
        return self

    @property
    def weights(self):
        """The model's TF variables by name.  Inference graph: what was restored.  Training graph: read back from the
        device, i.e. the CURRENT values after whatever train steps ran on this handle (rced_train_get_variables)."""
        if self.is_training and self._train is not None:
            buf = np.empty(spec.num_weights(self.variant), np.float32)
            _lib.check(_lib.load().rced_train_get_variables(self._train, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), buf.size))
            out, o = {}, 0
            for name, shape in spec.variable_shapes(self.variant):
                k = int(np.prod(shape))
                out[name] = buf[o:o + k].reshape(shape).copy()
                o += k
            return out
        return self._weights

    def check(self):
        """rced_check: after the caller has synchronised a device-resident `model(x)` itself, was that launch valid?
        Raises RcedError(RCED_ERR_STATE) if the fused kernel recorded a hand-off time-out (include/rced.h)."""
        if self._handle is not None:
            _lib.check(_lib.load().rced_check(self._handle))
        return True

    def param_count(self):
        """BaseTester.param_count (tester.py:41-47): trainable scalars."""
        return spec.num_trainable(self.variant)

    # -- options ---------------------------------------------------------------------------
    def set_path(self, path):
        """'auto' | 'layerwise' | 'fused' (rced.h RCED_PATH_*)."""
        code = {"auto": _lib.PATH_AUTO, "layerwise": _lib.PATH_LAYERWISE, "fused": _lib.PATH_FUSED}.get(path, path)
        self.set_option("path", int(code))
        return self

    def _infer_handle(self):
        if self._handle is None:
            raise RuntimeError("this option / query belongs to the inference graph (Model(is_training=False))")
        return self._handle

    def set_option(self, key, value):
        _lib.check(_lib.load().rced_set_option(self._infer_handle(), key.encode(), int(value)))
        self._options[key] = int(value)

    def get_option(self, key):
        v = ctypes.c_int()
        _lib.check(_lib.load().rced_get_option(self._infer_handle(), key.encode(), ctypes.byref(v)))
        return v.value

    def reserve(self, n, t):
        _lib.check(_lib.load().rced_reserve(self._infer_handle(), int(n), int(t)))

    def profile(self, on=True):
        self.set_option("profile", 1 if on else 0)

    def profile_query(self, kind):
        ms, cnt = ctypes.c_float(), ctypes.c_int()
        _lib.check(_lib.load().rced_profile_query(self._infer_handle(), int(kind), ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    # -- forward ---------------------------------------------------------------------------
    @staticmethod
    def _check_shape(shape):
        if len(shape) != 4 or shape[2] != spec.FEATURE_DIM or shape[3] != 1:
            raise ValueError("input must be [N, T, %d, 1] (NHWC), got %s" % (spec.FEATURE_DIM, tuple(shape)))

    def __call__(self, x, out=None):
        """y = model(x).  numpy in -> numpy out; torch.cuda in -> torch.cuda out.  `out`: an optional preallocated
        C-contiguous float32 buffer of x's shape to write into (a cuda tensor for a cuda input, an ndarray for an ndarray).
        Without `out` a numpy call returns a new ndarray object over recycled, already-touched pages (HostOutputPool): fresh
        as far as the caller can tell, without the ~5 ms of page faults a new 67 MB array costs at config 3."""
        lib = _lib.load()
        if _is_torch(x):
            import torch
            self._check_shape(x.shape)
            if not x.is_cuda:
                raise ValueError("torch input must be a CUDA/HIP tensor (pass numpy for host buffers)")
            if x.device.index != self.device:
                raise ValueError("input on cuda:%d, model on cuda:%d" % (x.device.index, self.device))
            if x.dtype != torch.float32:
                x = x.float()
            x = x.contiguous()
            if out is None:
                y = torch.empty_like(x)
            else:
                if (not _is_torch(out) or out.shape != x.shape or out.dtype != torch.float32 or out.device != x.device
                        or not out.is_contiguous()):
                    raise ValueError("out must be a contiguous float32 tensor of the input's shape on its device")
                y = out
            n, t = int(x.shape[0]), int(x.shape[1])
            if n and t:
                st = torch.cuda.current_stream(x.device).cuda_stream
                if self.is_training:
                    _lib.check(lib.rced_train_forward(self._train, x.data_ptr(), y.data_ptr(), n, t, st))
                else:
                    _lib.check(lib.rced_forward(self._handle, x.data_ptr(), y.data_ptr(), n, t, st))
            return y
        x = np.asarray(x)
        self._check_shape(x.shape)
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, t = x.shape[0], x.shape[1]
        if self.is_training:        # the training graph lives on device tensors: stage through torch
            import torch
            if not (n and t):
                return np.empty_like(x)
            y = self(torch.from_numpy(x).to("cuda:%d" % self.device)).cpu().numpy()
            if out is not None:
                np.copyto(out, y)
                return out
            return y
        if out is None:
            y = self._host_pool.take(x.shape) if self._host_pool is not None else np.empty_like(x)
        else:
            if (not isinstance(out, np.ndarray) or out.shape != x.shape or out.dtype != np.float32
                    or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]):
                raise ValueError("out must be a writeable C-contiguous float32 ndarray of the input's shape")
            y = out
        if n and t:
            _lib.check(lib.rced_forward_host(self._handle, x.ctypes.data, y.ctypes.data, n, t))
        return y

    # -- lifetime --------------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            try:
                _lib.load().rced_destroy(self._handle)
            except Exception:
                pass
            self._handle = None
        if self._train is not None:
            try:
                _lib.load().rced_train_destroy(self._train)
            except Exception:
                pass
            self._train = None

    def close(self):
        self._release()

    def __del__(self):
        self._release()


class FullyCNNSEModel(_RcedNet):
    """R-CED, 10 layers, 32,765 parameters (model.py:6-29, readme.md:65)."""
    variant = spec.V1


class FullyCNNSEModelV2(_RcedNet):
    """R-CED, 16 layers, 32,192 parameters (model.py:32-61, readme.md:66)."""
    variant = spec.V2


class FullyCNNSEModelV3(_RcedNet):
    """CR-CED, 16 layers, 32,653 parameters (model.py:64-96, readme.md:67)."""
    variant = spec.V3


def build_model(net_work, is_training=False, dtype="float32", **kw):
    """The selection block of tester.py:76-82 / infer.py:45-51.

    dtype="bfloat16" (R-CED V1 / V2 only; BASELINE config 2) switches the fused kernel to bf16 activations and
    weights -- outside the 1e-4 fp32 bar; the default is the reference's float32."""
    if net_work == "FullyCNNV2":
        m = FullyCNNSEModelV2(is_training, **kw)
    elif net_work == "FullyCNNV3":
        m = FullyCNNSEModelV3(is_training, **kw)
    else:
        m = FullyCNNSEModel(is_training, **kw)
    if dtype in ("bfloat16", "bf16"):
        m.set_option("bf16", 1)
    elif dtype not in ("float32", "f32", "fp32"):
        raise ValueError("dtype must be 'float32' or 'bfloat16', got %r" % (dtype,))
    return m


def conv_bn_relu(inputs, out_channels, kernel_size, stride=(1, 1), is_training=False, padding="SAME",
                 use_norm=True, use_act=True, scope="conv", skip_input=None, params=None):
    """The single op of model_utils/module.py:11-34 on the GPU (inference BN).

    inputs: torch.cuda float32 [N,T,F,cin]; params: dict with "{scope}/kernel", "{scope}/bias" and, if
    use_norm, "{scope}/batch_norm/{gamma,beta,moving_mean,moving_variance}" (numpy or torch).
    Returns a torch.cuda tensor [N,T,F,out_channels].

    is_training=True (module.py:29 `training=is_training`): BatchNorm normalises with the mean / biased variance of this
    batch (rced_conv_bn_relu_train); the moving statistics in `params` are neither read nor changed -- in TF they change
    only when the UPDATE_OPS run, which belongs to train_op (trainer.py:175-179).  With use_norm=False the flag has no
    effect, as in the reference.
    """
    import torch
    if tuple(stride) != (1, 1) or padding != "SAME":
        raise ValueError("only stride (1,1), padding 'SAME' (all the reference ever passes)")
    if params is None:
        raise ValueError("params (the layer's TF variables) are required")
    x = inputs
    if not (_is_torch(x) and x.is_cuda):
        raise ValueError("inputs must be a torch CUDA/HIP tensor")
    x = x.float().contiguous()
    n, t, f, cin = (int(s) for s in x.shape)
    kh, kw = int(kernel_size[0]), int(kernel_size[1])
    dev = x.device

    def dev_t(a, shape):
        a = torch.as_tensor(np.asarray(a.detach().cpu()) if _is_torch(a) else np.asarray(a), dtype=torch.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("parameter shape %s, expected %s" % (tuple(a.shape), tuple(shape)))
        return a.contiguous().to(dev)

    k = dev_t(params[scope + "/kernel"], (kh, kw, cin, out_channels))
    b = dev_t(params[scope + "/bias"], (out_channels,))
    bn = None
    if use_norm and is_training:
        p = scope + "/batch_norm/"
        gb = torch.cat([dev_t(params[p + v], (out_channels,)) for v in ("gamma", "beta")])
        skip = None
        if skip_input is not None:
            skip = skip_input.float().contiguous()
            if tuple(skip.shape) != (n, t, f, out_channels):
                raise ValueError("skip_input shape %s, expected %s" % (tuple(skip.shape), (n, t, f, out_channels)))
        y = torch.empty((n, t, f, out_channels), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().rced_conv_bn_relu_train(
            x.data_ptr(), y.data_ptr(), k.data_ptr(), b.data_ptr(), gb.data_ptr(),
            skip.data_ptr() if skip is not None else None, 1 if use_act else 0, n, t, f, cin, out_channels, kh, kw,
            None, dev.index, st))
        return y
    if use_norm:
        p = scope + "/batch_norm/"
        bn = torch.cat([dev_t(params[p + v], (out_channels,)) for v in ("gamma", "beta", "moving_mean", "moving_variance")])
    skip = None
    if skip_input is not None:
        skip = skip_input.float().contiguous()
        if tuple(skip.shape) != (n, t, f, out_channels):
            raise ValueError("skip_input shape %s, expected %s" % (tuple(skip.shape), (n, t, f, out_channels)))
    y = torch.empty((n, t, f, out_channels), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.load().rced_conv_bn_relu(
        x.data_ptr(), y.data_ptr(), k.data_ptr(), b.data_ptr(), bn.data_ptr() if bn is not None else None,
        skip.data_ptr() if skip is not None else None, 1 if use_act else 0, n, t, f, cin, out_channels, kh, kw,
        dev.index, st))
    return y
