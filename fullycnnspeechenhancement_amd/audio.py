"""Host-side mirror of the reference's STFT front-end and ISTFT rebuild, over the C ABI.

Reference surface kept:
    AudioFeature(windows_name).compute_spectrogram(signal, sample_rate, window_s, stride_s, nfft, use_complex)
        data_utils/audio_feature.py:12-44  (+ power_spectrum :102-110, divide_phase :113-115)
    AudioReBuild(windows_name, nfft).rebuild_audio(sig_length_list, spec, phase, sample_rate, windows_ms, stride_ms)
        model_utils/utils.py:93-183
Only the configuration the reference's cfgs use is built (8 kHz, 32 ms / 16 ms, hamming, rfft 256); anything
else raises.  The batch functions `stft_batch` / `istft_batch` keep everything on the device
(torch.cuda tensors in and out) so that STFT -> CNN -> ISTFT runs without leaving HBM.
"""

import numpy as np

from . import _lib

FRAME, STEP, BINS, SAMPLE_RATE = 256, 128, 129, 8000


def num_frames(length):
    return int(_lib.load().rced_stft_num_frames(int(length)))


KERNELS = {"x6": 1, "f32": 0}     # rced.h: RCED_AUDIO_X6 (the product: three-part bf16 products), RCED_AUDIO_F32 (the fp32-MFMA comparators)


def _kernels(name):
    if name not in KERNELS:
        raise ValueError("kernels must be 'x6' (the product) or 'f32' (the fp32-MFMA comparator), got %r" % (name,))
    return KERNELS[name]


def _check_cfg(sample_rate, window_s, stride_s, nfft=None):
    if int(round(window_s * sample_rate)) != FRAME or int(round(stride_s * sample_rate)) != STEP:
        raise ValueError("only 256-sample windows with a 128-sample stride are built (8 kHz, 32 ms / 16 ms)")
    if nfft is not None and nfft != 256:
        raise ValueError("only rfft(256) -> 129 bins is built (data_loader.py:59 hard-codes it)")


def stft_batch(pcm, lengths=None, frames=None, with_phase=True, kernels="x6"):
    """pcm: torch.cuda float32 [N, L]; lengths: per-utterance sample counts (list / tensor) or None.
    Returns (mag [N, T, 129, 1], phase [N, T, 129] complex64 or None); T = frames or the batch maximum
    (zero-padded like DataLoader.padding_batch, data_loader.py:198-209)."""
    import torch
    if not (pcm.is_cuda and pcm.dim() == 2):
        raise ValueError("pcm must be a CUDA/HIP tensor [N, L]")
    pcm = pcm.float().contiguous()
    n, L = int(pcm.shape[0]), int(pcm.shape[1])
    dev = pcm.device
    if lengths is None:
        lens = [L] * n
        ldev = None
    else:
        lens = [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
        if len(lens) != n or any(v < 1 or v > L for v in lens):
            raise ValueError("lengths must hold N values in [1, L]")
        ldev = torch.tensor(lens, dtype=torch.int32, device=dev)
    t = int(frames) if frames is not None else (max(num_frames(v) for v in lens) if n else 0)
    mag = torch.empty((n, t, BINS, 1), dtype=torch.float32, device=dev)
    ph = torch.empty((n, t, BINS, 2), dtype=torch.float32, device=dev) if with_phase else None
    if n and t:
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().rced_stft_ex(pcm.data_ptr(), ldev.data_ptr() if ldev is not None else None, n, L, t,
                                            mag.data_ptr(), ph.data_ptr() if ph is not None else None, dev.index, st, _kernels(kernels)))
    return mag, (torch.view_as_complex(ph) if ph is not None else None)


def istft_batch(mag, phase, nfft=512, kernels="x6"):
    """mag [N, T, 129(,1)] float32, phase [N, T, 129] complex64 (torch.cuda) -> audio [N, (T+1)*128]."""
    import torch
    if mag.dim() == 4:
        mag = mag.squeeze(-1)
    mag = mag.float().contiguous()
    ph = torch.view_as_real(phase.to(torch.complex64).contiguous()).contiguous()
    n, t = int(mag.shape[0]), int(mag.shape[1])
    if tuple(mag.shape) != (n, t, BINS) or tuple(ph.shape) != (n, t, BINS, 2):
        raise ValueError("mag must be [N, T, 129], phase [N, T, 129] complex")
    out = torch.empty((n, (t + 1) * STEP), dtype=torch.float32, device=mag.device)
    if n and t:
        st = torch.cuda.current_stream(mag.device).cuda_stream
        _lib.check(_lib.load().rced_istft_ex(mag.data_ptr(), ph.data_ptr(), n, t, int(nfft), out.data_ptr(),
                                             mag.device.index, st, _kernels(kernels)))
    return out


class AudioFeature(object):
    """data_utils/audio_feature.py:12-115 on the GPU (numpy in, numpy out, like the reference)."""

    def __init__(self, windows_name=None, device=0, kernels="x6"):
        if windows_name not in (None, "hamming"):
            raise ValueError("only the hamming window is built (it is what every reference run uses: SURVEY F7)")
        self.device, self.kernels = device, kernels
        _kernels(kernels)

    def compute_spectrogram(self, signal, sample_rate, window_s=0.02, stride_s=0.01, nfft=512, use_complex=False):
        import torch
        if stride_s > window_s:
            raise ValueError("Stride size must not be greater than window size.")   # audio_feature.py:29-30
        _check_cfg(sample_rate, window_s, stride_s, nfft)
        sig = torch.as_tensor(np.asarray(signal, dtype=np.float32), device="cuda:%d" % self.device)[None]
        mag, ph = stft_batch(sig, with_phase=use_complex, kernels=self.kernels)
        if use_complex:   # [129, T] complex, like np.transpose(fft_frames)
            return (mag[0, :, :, 0] * ph[0]).cpu().numpy().T
        return mag[0, :, :, 0].cpu().numpy().T

    @staticmethod
    def power_spectrum(frames):
        return np.absolute(frames)

    @staticmethod
    def divide_phase(fft_frames):
        return np.exp(1.j * np.angle(fft_frames))


class AudioReBuild(object):
    """model_utils/utils.py:93-183 on the GPU.  nfft defaults to 512 exactly as the reference's does."""

    def __init__(self, windows_name=None, nfft=512, device=0, kernels="x6"):
        if windows_name not in (None, "hamming"):
            raise ValueError("only the hamming window is built")
        if nfft not in (256, 512):
            raise ValueError("nfft must be 512 (reference default) or 256")
        self.nfft, self.device, self.kernels = nfft, device, kernels
        _kernels(kernels)

    def rebuild_audio(self, sig_length_list, spec, phase, sample_rate, windows_ms, stride_ms):
        import torch
        _check_cfg(sample_rate, windows_ms / 1000.0, stride_ms / 1000.0)
        dev = "cuda:%d" % self.device
        mag = torch.as_tensor(np.asarray(spec, dtype=np.float32), device=dev)
        ph = torch.as_tensor(np.asarray(phase).astype(np.complex64), device=dev)
        audio = istft_batch(mag, ph, self.nfft, kernels=self.kernels).cpu().numpy()
        return [audio[i][:sig_length_list[i]] for i in range(len(audio))]
