"""ORACLE (test infrastructure, NOT product code): numpy restatement of the STFT front-end and the
ISTFT rebuild that surround the CNN in the reference pipeline.

PARITY PINNED: unlike the CNN, these two pieces of the reference are numpy-only, so this restatement
is checked against outputs of the reference's OWN code (tests/golden/audio_stft.npz, produced by
tests/golden/make_golden_audio.py importing /root/reference/data_utils/audio_feature.py and
/root/reference/model_utils/utils.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

import numpy as np

PRE_EMPHASIS = 0.97     # audio_feature.py:53, utils.py:106
FRAME = 256             # round(0.032 * 8000): audio_feature.py:67
STEP = 128              # round(0.016 * 8000): audio_feature.py:68
NFFT = 256              # data_loader.py:59 (hard-coded)
BINS = 129


def num_frames(length):
    """audio_feature.py:70: ceil(|L - frame| / step + 1)."""
    return int(np.ceil(float(abs(int(length) - FRAME)) / STEP + 1))


def stft(sig):
    """AudioFeature.compute_spectrogram(use_complex=True) (audio_feature.py:22-44), transposed to [T, 129]:
    pre-emphasis (:47-55) -> framing with zero padding (:58-77) -> hamming (:79-88) -> rfft(256) (:91-99).
    Returns (magnitude [T,129] float64, unit phase [T,129] complex128) = power_spectrum / divide_phase."""
    # The reference pre-emphasises in the dtype librosa hands over (float32: the Python scalar 0.97 is
    # "weak"), i.e. one float32 multiply and one float32 subtract per sample; everything after the
    # zero-padding (np.append with float64 zeros, audio_feature.py:73) runs in float64.
    sig = np.asarray(sig, dtype=np.float32)
    emph = np.append(sig[0], sig[1:] - np.float32(PRE_EMPHASIS) * sig[:-1]).astype(np.float64)
    t = num_frames(len(sig))
    pad = np.zeros(t * STEP + FRAME)
    pad[:len(emph)] = emph
    frames = np.stack([pad[i * STEP:i * STEP + FRAME] for i in range(t)])
    frames = frames * np.hamming(FRAME)
    spec = np.fft.rfft(frames, NFFT)
    return np.abs(spec), np.exp(1j * np.angle(spec))


def rebuild(mag, phase, length=None, nfft=512):
    """AudioReBuild.rebuild_audio (utils.py:171-183) for one utterance: mag, phase [T, 129].
    merge (:119-126) -> irfft(n=nfft)[:256] (:115-117,176; the reference's default nfft is 512 although
    the STFT used 256 -- SURVEY F7; reproduced as shipped) -> divide by hamming (:128-137) -> keep the
    first half of frame 0 and the second half of every frame (:139-147) -> de-emphasis IIR (:104-113)."""
    x = np.asarray(mag, np.float64) * np.asarray(phase, np.complex128)
    fr = np.fft.irfft(x, nfft)[:, :FRAME] / np.hamming(FRAME)
    sig = np.append(fr[0, :FRAME - STEP], fr[:, FRAME - STEP:].reshape(-1))
    out = np.empty_like(sig)
    acc = 0.0
    for i, v in enumerate(sig):     # y[i] = x[i] + 0.97 * y[i-1]
        acc = v + PRE_EMPHASIS * acc if i else v
        out[i] = acc
    return out if length is None else out[:length]
