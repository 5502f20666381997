"""Host-side mirror of the reference's evaluation engines for the forward hot path.

Reference: BaseTester / FullyCNNTester (model_utils/tester.py:18-90) and InferenceEngine
(infer.py:19-52): read cfg -> creat_graph() -> _init_session() -> _load_checkpoint() ->
test_step(ndarray[N,T,129,1]) -> ndarray[N,T,129,1].  Only that slice is mirrored; the dataset
loop, ISTFT rebuild and PESQ/STOI/SDR scoring (tester.py:92-167) are outside the hot path.
"""

import numpy as np

from . import model as _model, spec, weights as _weights


class FullyCNNTester(object):
    """test_config: a configparser-like object (config.py:9-12) or None with keyword overrides.

    Keys read, as the reference does: [model] net_work (tester.py:21-22), [testing]
    checkpoint_filepath (tester.py:20) -- a TF V2 checkpoint prefix, a frozen .pb or a .npz of TF variables --, [data] feature_dim
    (tester.py:54).  The shipped infer cfgs name the section [inference]: both are accepted.
    """

    def __init__(self, test_config=None, net_work=None, checkpoint_file=None, weights=None, device=0, reuse_output=False):
        self.device = device
        # test_step's default already avoids the page faults of a fresh 67 MB ndarray per call (model.HostOutputPool: a new ndarray object
        # over recycled warm pages, never one the caller still holds).  reuse_output is the blunt form: one of TWO pooled arrays per shape,
        # alternating, whatever the caller holds -- the result of call k is overwritten by call k + 2.
        self.reuse_output = bool(reuse_output)
        self._out_pool = {}
        self.net_work = net_work
        self.checkpoint_file = checkpoint_file
        self.feature_dim = spec.FEATURE_DIM
        if test_config is not None:
            self.net_work = test_config.get("model", "net_work")
            for section in ("testing", "inference"):
                if test_config.has_section(section) and test_config.has_option(section, "checkpoint_filepath"):
                    self.checkpoint_file = test_config.get(section, "checkpoint_filepath")
                    break
            if test_config.has_option("data", "feature_dim"):
                self.feature_dim = int(test_config.get("data", "feature_dim"))
        if self.feature_dim != spec.FEATURE_DIM:
            raise ValueError("feature_dim must be %d (nfft 256), got %d" % (spec.FEATURE_DIM, self.feature_dim))
        self._weights = weights
        self.creat_graph()
        self._load_checkpoint()

    def creat_graph(self):
        """tester.py:69-83: pick the model by net_work and build pred = model(input_x)."""
        if self.net_work not in ("FullyCNNV2", "FullyCNNV3"):
            print("net_work set default or not wright. Use FullyCNN")
        self.model = _model.build_model(self.net_work, is_training=False, device=self.device)

    def _load_checkpoint(self):
        """tester.py:36-39."""
        if self._weights is not None:
            self.model.restore(self._weights)
        elif self.checkpoint_file:
            self.model.restore(_weights.load(self.checkpoint_file, self.model.variant))
            print("recover from checkpoint_file: {}".format(self.checkpoint_file))

    def param_count(self):
        """tester.py:41-47."""
        n = self.model.param_count()
        print("\nTotal number of Parameters: {}\n".format(n))
        return n

    def test_step(self, input_x, out=None):
        """tester.py:85-90: output = sess.run(self.pred, {self.input_x: input_x}).
        `out`: a caller-owned C-contiguous float32 ndarray of the input's shape to write into; with reuse_output=True (and no
        `out`) the result lands in one of two pooled arrays of this shape, alternating."""
        if out is None and self.reuse_output and isinstance(input_x, np.ndarray):
            key = tuple(input_x.shape)
            pool = self._out_pool.get(key)
            if pool is None:
                pool = self._out_pool[key] = [np.empty(key, np.float32), np.empty(key, np.float32), 0]
                if len(self._out_pool) > 4:                       # a few shapes at most: drop the oldest
                    self._out_pool.pop(next(iter(self._out_pool)))
            out = pool[pool[2]]
            pool[2] ^= 1
        return self.model(input_x, out=out) if out is not None else self.model(input_x)


class InferenceEngine(FullyCNNTester):
    """infer.py:19-52, the forward slice: `denoise_magnitude` is infer.py:62-65 without the
    STFT/ISTFT around it."""

    def denoise_pcm(self, sig, nfft=512):
        """infer.py:54-71 end to end on the device: STFT (audio_feature.py) -> model -> rebuild (utils.py:171-183).
        The magnitude is laid out [1, T, 129, 1] by TRANSPOSE, as the batch loader does
        (data_loader.py:206-208); infer.py:59 itself reshapes without transposing.
        sig: 1-D float PCM at 8 kHz.  Returns the denoised signal, same length, numpy float32."""
        import torch
        from . import audio
        dev = "cuda:%d" % self.device
        pcm = torch.as_tensor(np.asarray(sig, dtype=np.float32), device=dev)[None]
        mag, phase = audio.stft_batch(pcm)
        pred = self.model(mag)
        out = audio.istft_batch(pred, phase, nfft)
        return out[0, :len(sig)].cpu().numpy()

    def denoise_magnitude(self, mag):
        mag = np.asarray(mag, dtype=np.float32)
        if mag.ndim == 2:  # [T, 129] -> [1, T, 129, 1]  (a transpose-correct version of infer.py:59)
            mag = mag[None, :, :, None]
        return self.test_step(mag)
