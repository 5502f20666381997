"""Host-side mirror of the reference's training step, over the C ABI (rced_train_*).

Reference: FullyCNNTrainer (model_utils/trainer.py:121-192) -- creat_graph builds
`pred = Model(is_training=True)(input_x)`, `loss = sum((target - pred)^2) / batch_size`, Adam under the
BatchNorm UPDATE_OPS; train_step runs `[loss, summaries, global_step, train_op]`; the training loop feeds the
learning rate: `init_lr` for the first step, then the Noam schedule of the returned global step
(trainer.py:27,68-76,215).  Loops, checkpoints and TF summaries are host orchestration and not mirrored.
"""

import ctypes

import numpy as np

from . import _lib, spec, weights as _weights


class FullyCNNTrainer(object):
    def __init__(self, net_work="FullyCNNV3", batch_size=1, lr=1e-3, warmup_steps=4000.0, weights=None, device=0,
                 seed=None):
        self.net_work = net_work
        self.variant = spec.variant_of(net_work)
        self.batch_size = int(batch_size)          # [training] batch_size: what the loss divides by
        self.init_lr = float(lr)                   # [training] lr
        self.lr = float(lr)                        # fed for the next step (trainer.py:27)
        self.warmup_steps = float(warmup_steps)    # [training] warmup_steps
        self.device = int(device)
        w = weights if weights is not None else _weights.initial_weights(self.variant, seed)
        self._blob_n = spec.num_weights(self.variant)
        blob = _weights.pack_blob(self.variant, w)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().rced_train_create(self.variant, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                 blob.size, self.batch_size, self.device, ctypes.byref(self._h)))

    def noam_scheme(self, global_step, warmup_steps=None):
        """trainer.py:68-76."""
        w = self.warmup_steps if warmup_steps is None else warmup_steps
        step = global_step + 1
        return self.init_lr * w ** 0.5 * min(step * w ** -1.5, step ** -0.5)

    def train_step(self, input_x, target_y):
        """trainer.py:181-192: returns (batch_loss, summary (None here), global_step)."""
        import torch
        dev = "cuda:%d" % self.device
        x = torch.as_tensor(np.asarray(input_x, dtype=np.float32) if not hasattr(input_x, "is_cuda") else input_x,
                            device=dev).float().contiguous()
        y = torch.as_tensor(np.asarray(target_y, dtype=np.float32) if not hasattr(target_y, "is_cuda") else target_y,
                            device=dev).float().contiguous()
        if x.dim() != 4 or x.shape[2] != spec.FEATURE_DIM or x.shape[3] != 1 or x.shape != y.shape:
            raise ValueError("input and target must both be [N, T, 129, 1]")
        loss = ctypes.c_double()
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.load().rced_train_step(self._h, x.data_ptr(), y.data_ptr(), int(x.shape[0]), int(x.shape[1]),
                                               ctypes.c_float(self.lr), ctypes.byref(loss), st))
        return loss.value, None, self.global_step

    def fit_step(self, input_x, target_y):
        """One iteration of the loop body of trainer.py:212-215: step, then set lr for the next step."""
        out = self.train_step(input_x, target_y)
        self.lr = self.noam_scheme(out[2], self.warmup_steps)
        return out

    @property
    def global_step(self):
        return int(_lib.load().rced_train_global_step(self._h))

    def _blob(self, fn):
        buf = np.empty(self._blob_n, np.float32)
        _lib.check(fn(self._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), buf.size))
        out, o = {}, 0
        for name, shape in spec.variable_shapes(self.variant):
            n = int(np.prod(shape))
            out[name] = buf[o:o + n].reshape(shape).copy()
            o += n
        return out

    def variables(self):
        """Current TF variables (incl. moving statistics), keyed by name -- what Saver.save would write."""
        return self._blob(_lib.load().rced_train_get_variables)

    def gradients(self):
        return self._blob(_lib.load().rced_train_get_gradients)

    def save_checkpoint(self, prefix):
        """Write the model variables + global_step as a TF V2 checkpoint (trainer.py:98-104 naming is the
        caller's).  The reference's test / infer / freeze graphs restore exactly these variables
        (tester.py:36-39); Adam slots are not exported, so the reference *trainer* cannot continue from it."""
        from . import tf_checkpoint
        tensors = dict(self.variables())
        tensors["global_step"] = np.asarray(self.global_step, np.int64)
        tf_checkpoint.write_checkpoint(prefix, tensors)
        return prefix

    def close(self):
        if self._h is not None:
            _lib.load().rced_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
