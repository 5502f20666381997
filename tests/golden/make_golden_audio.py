"""Generates tests/golden/audio_stft.npz from the REFERENCE'S OWN numpy code, run in this container:
    data_utils/audio_feature.py  AudioFeature.compute_spectrogram / divide_phase   (STFT front-end)
    model_utils/utils.py         AudioReBuild.rebuild_audio                        (ISTFT rebuild)
These two classes are numpy-only, so -- unlike the TensorFlow CNN -- their outputs are real reference
outputs and PIN the STFT / ISTFT rows.  Nothing of the reference's source is copied: the script imports
it from /root/reference, feeds seeded random PCM and stores inputs + outputs.

Two environment shims, neither touching the algorithm:
  * `np.mat` was removed in NumPy 2.0; the reference calls it in en_frame (audio_feature.py:76) only to
    build an index array -> aliased to np.asmatrix (its documented replacement);
  * model_utils/utils.py imports librosa / pypesq / pystoi at module level (utils.py:7-10) for its PESQ /
    STOI / SDR classes; they are not installed and AudioReBuild never touches them -> empty stand-in
    modules are registered just so that the file imports.

Run from the repo root:  python tests/golden/make_golden_audio.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix
    for name, attrs in (("librosa", ()), ("pypesq", ("pesq",)), ("pystoi", ("stoi",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    from data_utils.audio_feature import AudioFeature
    from model_utils.utils import AudioReBuild

    sr, window_ms, stride_ms = 8000, 32, 16                 # Work/*/cfg: sample_rate 8000, 32 / 16 ms
    rng = np.random.default_rng(2024)
    out = {}
    lengths = [4000, 2000, 1234, 256, 300, 100, 8192]
    out["lengths"] = np.asarray(lengths, np.int32)
    fe, rb = AudioFeature(), AudioReBuild()                  # hamming window; nfft=512 rebuild default
    for i, L in enumerate(lengths):
        t = np.arange(L) / sr
        sig = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * rng.standard_normal(L)).astype(np.float32)
        spec = fe.compute_spectrogram(sig, sr, window_s=window_ms / 1000, stride_s=stride_ms / 1000, nfft=256,
                                      use_complex=True)     # [129, T] complex128 (data_loader.py:54-61)
        mag = fe.power_spectrum(spec)                        # infer.py:58
        phase = fe.divide_phase(spec)                        # infer.py:60
        mag_f32 = fe.compute_spectrogram(sig, sr, window_s=window_ms / 1000, stride_s=stride_ms / 1000, nfft=256,
                                         use_complex=False)  # the training-side float32 magnitude
        magT, phaseT = mag.T[None], phase.T[None]            # [1, T, 129] as padding_batch lays them out
        audio = rb.rebuild_audio([L], magT, phaseT, sr, window_ms, stride_ms)[0]
        out["pcm_%d" % i] = sig
        out["mag_%d" % i] = mag.T.astype(np.float64)        # [T, 129]
        out["mag32_%d" % i] = mag_f32.T                      # [T, 129] float32 as the reference returns it
        out["phase_%d" % i] = phaseT[0].astype(np.complex128)
        out["audio_%d" % i] = np.asarray(audio, np.float64)  # rebuilt signal, trimmed to L
    np.savez_compressed(os.path.join(HERE, "audio_stft.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
