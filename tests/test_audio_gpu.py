"""GPU parity of the STFT front-end / ISTFT rebuild kernels, through the C ABI, against REAL reference
outputs (tests/golden/audio_stft.npz: the reference's own AudioFeature / AudioReBuild run on seeded PCM).
Tolerances: the kernels are fp32 GEMMs (K = 256 / 258); the reference computes in float64."""

import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import audio_np

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=["x6", "fp32-mfma"], autouse=True)
def audio_kernels(request, built):
    """Every test of this file runs with both kernel families: the three-part bf16 kernels (the default) and the fp32-MFMA
    comparators (rced_audio_option "x6" = 0) -- same reference-pinned tolerances."""
    from fullycnnspeechenhancement_amd import audio
    prev = audio.kernel_option("x6")
    assert audio.kernel_option("x6", request.param) == request.param
    yield request.param
    audio.kernel_option("x6", prev)


def test_audio_option_contract(built):
    from fullycnnspeechenhancement_amd import _lib
    lib = _lib.load()
    now = lib.rced_audio_option(b"x6", -1)
    assert now in (0, 1) and lib.rced_audio_option(b"x6", 7) == -1 and lib.rced_audio_option(b"nonsense", 1) == -1
    assert lib.rced_audio_option(b"x6", -1) == now            # a refused call changes nothing


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(ROOT, "tests", "golden", "audio_stft.npz"))
    return {k: z[k] for k in z.files}


def test_stft_single_signals_match_reference(gold, built):
    from fullycnnspeechenhancement_amd.audio import AudioFeature, num_frames
    fe = AudioFeature()
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


def test_stft_ragged_batch_is_zero_padded_like_the_loader(gold, built):
    import torch
    from fullycnnspeechenhancement_amd.audio import stft_batch
    lens = [int(v) for v in gold["lengths"]]
    L = max(lens)
    pcm = np.zeros((len(lens), L), np.float32)
    for i, n in enumerate(lens):
        pcm[i, :n] = gold["pcm_%d" % i]
        pcm[i, n:] = 7.0                      # junk past the utterance must not leak in
    mag, ph = stft_batch(torch.from_numpy(pcm).cuda(), lens)
    T = max(gold["mag_%d" % i].shape[0] for i in range(len(lens)))
    assert tuple(mag.shape) == (len(lens), T, 129, 1) and tuple(ph.shape) == (len(lens), T, 129)
    mag, ph = mag.cpu().numpy()[..., 0], ph.cpu().numpy()
    for i in range(len(lens)):
        ref = gold["mag_%d" % i]
        t = ref.shape[0]
        assert np.abs(mag[i, :t] - ref).max() <= 2e-6 * ref.max()
        assert not mag[i, t:].any()                                   # padding_batch zero-fills
        assert np.all(ph[i, t:] == 1.0 + 0.0j)                        # exp(j*angle(0)) = 1


def test_rebuild_matches_reference(gold, built):
    from fullycnnspeechenhancement_amd.audio import AudioReBuild
    rb = AudioReBuild()                        # nfft = 512, the reference's shipped default
    for i, L in enumerate(gold["lengths"]):
        out = rb.rebuild_audio([int(L)], gold["mag_%d" % i][None], gold["phase_%d" % i][None], 8000, 32, 16)[0]
        ref = gold["audio_%d" % i]
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max()


def test_stft_then_matching_inverse_round_trip(gold, built):
    """Size-independent property at a long length: nfft = 256 inverts the front-end."""
    import torch
    from fullycnnspeechenhancement_amd.audio import istft_batch, stft_batch
    rng = np.random.default_rng(5)
    sig = (0.2 * rng.standard_normal((3, 65536))).astype(np.float32)
    x = torch.from_numpy(sig).cuda()
    mag, ph = stft_batch(x)
    back = istft_batch(mag, ph, nfft=256)[:, :sig.shape[1]].cpu().numpy()
    assert np.abs(back - sig).max() <= 2e-4 * np.abs(sig).max()
    shipped = istft_batch(mag, ph, nfft=512)[:, :sig.shape[1]].cpu().numpy()
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


def test_rebuild_fused_and_split_paths_agree(built):
    """The three-part ISTFT kernel has two shapes (audio_api.hip): with enough utterances one workgroup walks a whole utterance and
    does de_frame and de_emphasis itself (a blocked scan with the carry across its 64-frame blocks); a small batch is cut into frame
    ranges over more workgroups and the head / de_emphasis kernels follow.  Same spectra through both: equal to fp32 scan noise, and
    both against the numpy restatement of the reference's rebuild."""
    import torch
    from fullycnnspeechenhancement_amd.audio import istft_batch, stft_batch
    rng = np.random.default_rng(11)
    sig = (0.2 * rng.standard_normal((3, 20000))).astype(np.float32)     # T = 156 frames: three 64-frame blocks
    mag, ph = stft_batch(torch.from_numpy(sig).cuda())
    small = istft_batch(mag, ph, nfft=512).cpu().numpy()                 # 3 utterances: split over frame ranges
    big = istft_batch(mag.repeat(100, 1, 1, 1), ph.repeat(100, 1, 1), nfft=512).cpu().numpy()   # 300 utterances: one workgroup each
    assert np.array_equal(big[:3], big[297:]) and np.array_equal(big[:3], big[150:153])
    scale = np.abs(small).max()
    assert np.abs(big[:3] - small).max() <= 2e-6 * scale
    ref = audio_np.rebuild(*audio_np.stft(sig[2]), length=sig.shape[1], nfft=512)
    assert np.abs(big[2, :sig.shape[1]] - ref).max() <= 5e-5 * np.abs(ref).max()
