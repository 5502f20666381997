"""GPU parity of the STFT front-end / ISTFT rebuild kernels, through the C ABI, against REAL reference
outputs (tests/golden/audio_stft.npz: the reference's own AudioFeature / AudioReBuild run on seeded PCM).
Tolerances: the kernels are fp32 GEMMs (K = 256 / 258); the reference computes in float64."""

import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import audio_np

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["x6", "f32"], ids=["x6", "fp32-mfma"])
def K(request, built):
    """Every parity test of this file runs with both kernel families -- the three-part bf16 kernels (what rced_stft / rced_istft
    launch) and the fp32-MFMA comparators -- selected per CALL (rced_stft_ex / rced_istft_ex's last argument): same
    reference-pinned tolerances."""
    return request.param


def test_kernel_family_is_an_argument(built):
    """No process-wide switch: rced_stft / rced_istft are the product kernels, the _ex entries take the family; a value that is
    neither RCED_AUDIO_X6 nor RCED_AUDIO_F32 is refused."""
    import torch
    from fullycnnspeechenhancement_amd import _lib, audio
    lib = _lib.load()
    assert not hasattr(audio, "kernel_option")
    pcm = torch.randn(2, 2048, device="cuda")
    mag = torch.empty(2, 15, 129, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.rced_stft_ex(pcm.data_ptr(), None, 2, 2048, 15, mag.data_ptr(), None, 0, st, 7) == _lib.RCED_ERR_ARG
    assert lib.rced_stft_ex(pcm.data_ptr(), None, 2, 2048, 15, mag.data_ptr(), None, 0, st, 1) == 0
    m1 = mag.clone()
    assert lib.rced_stft(pcm.data_ptr(), None, 2, 2048, 15, mag.data_ptr(), None, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(m1, mag)                               # rced_stft IS the x6 family
    with pytest.raises(ValueError):
        audio.stft_batch(pcm, kernels="fast")


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(ROOT, "tests", "golden", "audio_stft.npz"))
    return {k: z[k] for k in z.files}


def test_stft_single_signals_match_reference(gold, K):
    from fullycnnspeechenhancement_amd.audio import AudioFeature, num_frames
    fe = AudioFeature(kernels=K)
    for i, L in enumerate(gold["lengths"]):
        assert num_frames(L) == gold["mag_%d" % i].shape[0]
        spec = fe.compute_spectrogram(gold["pcm_%d" % i], 8000, 0.032, 0.016, 256, use_complex=True)   # [129, T]
        mag, ref = np.abs(spec).T, gold["mag_%d" % i]
        scale = ref.max()
        assert mag.shape == ref.shape and np.abs(mag - ref).max() <= 2e-6 * scale
        ph, pref = fe.divide_phase(spec).T, gold["phase_%d" % i]
        strong = ref > 1e-3 * scale            # the phase of a near-zero bin is ill-conditioned
        assert np.abs(ph - pref)[strong].max() <= 2e-3
        m32 = fe.compute_spectrogram(gold["pcm_%d" % i], 8000, 0.032, 0.016, 256, use_complex=False)
        assert m32.dtype == np.float32 and np.abs(m32.T - gold["mag32_%d" % i]).max() <= 2e-6 * scale


def test_stft_ragged_batch_is_zero_padded_like_the_loader(gold, K):
    import torch
    from fullycnnspeechenhancement_amd.audio import stft_batch
    lens = [int(v) for v in gold["lengths"]]
    L = max(lens)
    pcm = np.zeros((len(lens), L), np.float32)
    for i, n in enumerate(lens):
        pcm[i, :n] = gold["pcm_%d" % i]
        pcm[i, n:] = 7.0                      # junk past the utterance must not leak in
    mag, ph = stft_batch(torch.from_numpy(pcm).cuda(), lens, kernels=K)
    T = max(gold["mag_%d" % i].shape[0] for i in range(len(lens)))
    assert tuple(mag.shape) == (len(lens), T, 129, 1) and tuple(ph.shape) == (len(lens), T, 129)
    mag, ph = mag.cpu().numpy()[..., 0], ph.cpu().numpy()
    for i in range(len(lens)):
        ref = gold["mag_%d" % i]
        t = ref.shape[0]
        assert np.abs(mag[i, :t] - ref).max() <= 2e-6 * ref.max()
        assert not mag[i, t:].any()                                   # padding_batch zero-fills
        assert np.all(ph[i, t:] == 1.0 + 0.0j)                        # exp(j*angle(0)) = 1


def test_rebuild_matches_reference(gold, K):
    from fullycnnspeechenhancement_amd.audio import AudioReBuild
    rb = AudioReBuild(kernels=K)               # nfft = 512, the reference's shipped default
    for i, L in enumerate(gold["lengths"]):
        out = rb.rebuild_audio([int(L)], gold["mag_%d" % i][None], gold["phase_%d" % i][None], 8000, 32, 16)[0]
        ref = gold["audio_%d" % i]
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max()


def test_stft_then_matching_inverse_round_trip(gold, K):
    """Size-independent property at a long length: nfft = 256 inverts the front-end."""
    import torch
    from fullycnnspeechenhancement_amd.audio import istft_batch, stft_batch
    rng = np.random.default_rng(5)
    sig = (0.2 * rng.standard_normal((3, 65536))).astype(np.float32)
    x = torch.from_numpy(sig).cuda()
    mag, ph = stft_batch(x, kernels=K)
    back = istft_batch(mag, ph, nfft=256, kernels=K)[:, :sig.shape[1]].cpu().numpy()
    assert np.abs(back - sig).max() <= 2e-4 * np.abs(sig).max()
    shipped = istft_batch(mag, ph, nfft=512, kernels=K)[:, :sig.shape[1]].cpu().numpy()
    ref = audio_np.rebuild(*audio_np.stft(sig[1]), length=sig.shape[1], nfft=512)
    assert np.abs(shipped[1] - ref).max() <= 5e-5 * np.abs(ref).max()


def test_pipeline_denoise_pcm_matches_oracle_chain(gold, built):
    """infer.py's STFT -> CNN -> ISTFT chain on the device against the same chain in the oracles."""
    from fullycnnspeechenhancement_amd import InferenceEngine
    from oracle import rced_c, rced_np
    w = rced_np.make_weights("FullyCNNV3", seed=42)
    eng = InferenceEngine(net_work="FullyCNNV3", weights=w)
    sig = gold["pcm_0"]
    out = eng.denoise_pcm(sig)
    mag, phase = audio_np.stft(sig)
    pred = rced_c.forward("FullyCNNV3", w, mag.astype(np.float32)[None, :, :, None], np.float64)[0, :, :, 0]
    ref = audio_np.rebuild(pred, phase, len(sig))
    assert out.shape == sig.shape and np.abs(out - ref).max() <= 1e-4 * np.abs(ref).max()


def test_rebuild_fused_and_split_paths_agree(K):
    """The three-part ISTFT kernel has two shapes (audio_api.hip): with enough utterances one workgroup walks a whole utterance and
    does de_frame and de_emphasis itself (a blocked scan with the carry across its 64-frame blocks); a small batch is cut into frame
    ranges over more workgroups and the head / de_emphasis kernels follow.  Same spectra through both: equal to fp32 scan noise, and
    both against the numpy restatement of the reference's rebuild."""
    import torch
    from fullycnnspeechenhancement_amd.audio import istft_batch, stft_batch
    rng = np.random.default_rng(11)
    sig = (0.2 * rng.standard_normal((3, 20000))).astype(np.float32)     # T = 156 frames: three 64-frame blocks
    mag, ph = stft_batch(torch.from_numpy(sig).cuda(), kernels=K)
    small = istft_batch(mag, ph, nfft=512, kernels=K).cpu().numpy()                 # 3 utterances: split over frame ranges
    big = istft_batch(mag.repeat(100, 1, 1, 1), ph.repeat(100, 1, 1), nfft=512, kernels=K).cpu().numpy()   # 300 utterances: one workgroup each
    assert np.array_equal(big[:3], big[297:]) and np.array_equal(big[:3], big[150:153])
    scale = np.abs(small).max()
    assert np.abs(big[:3] - small).max() <= 2e-6 * scale
    ref = audio_np.rebuild(*audio_np.stft(sig[2]), length=sig.shape[1], nfft=512)
    assert np.abs(big[2, :sig.shape[1]] - ref).max() <= 5e-5 * np.abs(ref).max()
