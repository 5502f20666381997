"""Host-side mirror of the reference's training step, over the C ABI (rced_train_*).

Reference: FullyCNNTrainer (model_utils/trainer.py:121-192) -- creat_graph builds
`pred = Model(is_training=True)(input_x)`, `loss = sum((target - pred)^2) / batch_size`, Adam under the
BatchNorm UPDATE_OPS; train_step runs `[loss, summaries, global_step, train_op]`; the training loop feeds the
learning rate: `init_lr` for the first step, then the Noam schedule of the returned global step
(trainer.py:27,68-76,215).  Loops, checkpoints and TF summaries are host orchestration and not mirrored.
"""

import ctypes

import numpy as np

from . import _lib, model as _model, spec, weights as _weights


class FullyCNNTrainer(object):
    model = None

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
        # creat_graph (trainer.py:165-172): self.model = Model(is_training=True); self.pred = self.model(self.input_x).
        # The model object owns the library's training handle; train_step runs on the same one.
        self.model = _model.build_model(net_work, True, weights=w, device=self.device, batch_size=self.batch_size)

    @property
    def _h(self):
        """The library's training handle, looked up on every call: the model object owns it, and
        `trainer.model.restore(w)` / `trainer.model.close()` replace / free it (a cached pointer would dangle)."""
        h = self.model._train if self.model is not None else None
        if h is None:
            raise RuntimeError("the trainer's model is closed")
        return h

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

    def valid_step(self, input_x):
        """trainer.py:245-250: `sess.run(self.pred)` on the training graph.  That graph was built with
        is_training=True (trainer.py:165-172), so BatchNorm uses the statistics of the batch it is given; fetching
        only `pred` runs no UPDATE_OPS and no optimizer step, so nothing changes.  ndarray in -> ndarray out, cuda
        tensor in -> cuda tensor out.  (The inference graph of tester.py / infer.py, which normalises with the moving
        statistics, is `build_model(net_work, False, weights=trainer.variables())`.)"""
        return self.model(input_x)

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

    # -- optimizer state / checkpoints (trainer.py:50-65, 232-239) ------------------------------------
    def optimizer_state(self):
        """(adam_m, adam_v, global_step): dicts keyed by TRAINABLE variable name (TF slots `<var>/Adam`, `<var>/Adam_1`)."""
        n = self._blob_n
        mb, vb = np.empty(n, np.float32), np.empty(n, np.float32)
        step = ctypes.c_longlong()
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.check(_lib.load().rced_train_get_state(self._h, mb.ctypes.data_as(fp), vb.ctypes.data_as(fp), n, ctypes.byref(step)))
        m, v, o = {}, {}, 0
        for name, shape in spec.variable_shapes(self.variant):
            k = int(np.prod(shape))
            if "moving_" not in name:
                m[name], v[name] = mb[o:o + k].reshape(shape).copy(), vb[o:o + k].reshape(shape).copy()
            o += k
        return m, v, int(step.value)

    def load_optimizer_state(self, adam_m, adam_v, global_step):
        """Resume: set the Adam moments (dicts as returned by optimizer_state; missing names start at zero, as a
        variable without slots would in TF) and the step counter; the next learning rate follows the Noam schedule."""
        n = self._blob_n
        mb, vb, o = np.zeros(n, np.float32), np.zeros(n, np.float32), 0
        for name, shape in spec.variable_shapes(self.variant):
            k = int(np.prod(shape))
            for blob, src in ((mb, adam_m), (vb, adam_v)):
                if src is not None and name in src:
                    a = np.asarray(src[name], np.float32)
                    if a.shape != tuple(shape):
                        raise ValueError("optimizer slot of %r has shape %s, expected %s" % (name, a.shape, tuple(shape)))
                    blob[o:o + k] = a.reshape(-1)
            o += k
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.check(_lib.load().rced_train_set_state(self._h, mb.ctypes.data_as(fp), vb.ctypes.data_as(fp), n, int(global_step)))
        if global_step > 0:
            self.lr = self.noam_scheme(int(global_step), self.warmup_steps)     # what the loop would have set (trainer.py:215)

    def save_checkpoint(self, prefix, with_optimizer=True):
        """Write a TF V2 checkpoint holding what `tf.train.Saver(tf.global_variables())` stores for the reference's
        training graph (trainer.py:50-51): the model variables, `global_step`, and -- with_optimizer -- the Adam
        slots `<var>/Adam`, `<var>/Adam_1` and the `beta1_power` / `beta2_power` accumulators (beta^(t+1) after t
        steps, as tf.train.AdamOptimizer keeps them).  The reference's test / infer / freeze graphs restore the model
        variables from it (tester.py:36-39); `FullyCNNTrainer.from_checkpoint` resumes from it.  Also written: the
        `checkpoint` state file `tf.train.latest_checkpoint` needs.  NOT written: `<prefix>.meta` (a serialized
        MetaGraphDef of the TF graph, which only TensorFlow can produce) -- the reference's own `continue_train`
        restores only when `continue_from + '.meta'` exists (trainer.py:59-65), so resuming THERE additionally needs
        a .meta from any checkpoint of the same graph (or `touch`, since only its existence is tested)."""
        from . import tf_checkpoint
        tensors = dict(self.variables())
        m, v, step = self.optimizer_state()
        tensors["global_step"] = np.asarray(step, np.int64)
        if with_optimizer:
            for name in m:
                tensors[name + "/Adam"] = m[name]
                tensors[name + "/Adam_1"] = v[name]
            tensors["beta1_power"] = np.asarray(0.9 ** (step + 1), np.float32)
            tensors["beta2_power"] = np.asarray(0.999 ** (step + 1), np.float32)
        tf_checkpoint.write_checkpoint(prefix, tensors)
        tf_checkpoint.write_checkpoint_state(prefix)
        return prefix

    @classmethod
    def from_checkpoint(cls, path, net_work="FullyCNNV3", **kw):
        """trainer.py:52-65 `continue_train`: variables, Adam slots (if the checkpoint has them) and global_step."""
        from . import tf_checkpoint
        variant = spec.variant_of(net_work)
        weights, adam_m, adam_v, step = tf_checkpoint.load_training_state(path, variant)
        tr = cls(net_work, weights=weights, **kw)
        tr.load_optimizer_state(adam_m, adam_v, step)
        return tr

    def restore(self, weights, keep_optimizer=True):
        """Saver.restore of model variables into a running trainer (trainer.py:59-65).  keep_optimizer: the Adam slots and
        global_step survive (a TF restore of a checkpoint that holds only model variables leaves them as they are);
        False resets them, which is what `trainer.model.restore(weights)` alone does."""
        state = self.optimizer_state() if keep_optimizer else None
        self.model.restore(weights)
        if state is not None:
            self.load_optimizer_state(*state)
        return self

    def close(self):
        if self.model is not None:
            self.model.close()     # the model owns the handle

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
