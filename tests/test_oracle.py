"""CPU tests of the oracle itself: the only pins available for an unpinned-parity path.

(1) the reference's published parameter counts (readme.md:65-67), (2) analytic known answers for
the TF semantics the restatement hard-codes (SAME asymmetry, BN eps, skip/ReLU order), (3) three
independent restatements agreeing, (4) the committed golden vectors.
"""

import numpy as np
import pytest

from conftest import NETS, load_golden, rel_err
from oracle import layers as L, rced_c, rced_np


@pytest.mark.parametrize("net_work,count,nlayers", [("FullyCNN", 32765, 10), ("FullyCNNV2", 32192, 16),
                                                    ("FullyCNNV3", 32653, 16), ("anything-else", 32765, 10)])
def test_param_counts_match_readme(net_work, count, nlayers):
    layers = L.layers_for(net_work)
    assert len(layers) == nlayers
    assert L.param_count(layers) == count


def test_same_padding_split():
    assert rced_np.same_pad(8) == (3, 4)      # kh = 8: 3 past frames, 4 future frames
    assert rced_np.same_pad(9) == (4, 4)
    assert rced_np.same_pad(129) == (64, 64)
    assert rced_np.same_pad(1) == (0, 0)


def test_delta_kernel_is_identity():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 129, 3))
    for kh, kw in ((1, 9), (8, 9), (1, 5), (1, 129)):
        k = np.zeros((kh, kw, 3, 3))
        pt, pl = rced_np.same_pad(kh)[0], rced_np.same_pad(kw)[0]
        for c in range(3):
            k[pt, pl, c, c] = 1.0
        y = rced_np.conv2d_same(x, k, np.zeros(3))
        np.testing.assert_allclose(y, x, atol=1e-12)


@pytest.mark.parametrize("kw", [5, 7, 9, 11, 13, 129])
def test_ones_kernel_counts_valid_taps_in_frequency(kw):
    x = np.ones((1, 2, 129, 1))
    y = rced_np.conv2d_same(x, np.ones((1, kw, 1, 1)), np.zeros(1))[0, 0, :, 0]
    pl, pr = rced_np.same_pad(kw)
    f = np.arange(129)
    expect = np.minimum(f + pr, 128) - np.maximum(f - pl, 0) + 1
    np.testing.assert_allclose(y, expect)


def test_kh8_sees_three_past_and_four_future_frames():
    T = 12
    x = np.zeros((1, T, 129, 1))
    x[0, 5] = 1.0                       # one hot frame
    k = np.zeros((8, 1, 1, 1))
    k[:, 0, 0, 0] = np.arange(1, 9)     # tap i has weight i+1
    y = rced_np.conv2d_same(x, k, np.zeros(1))[0, :, 0, 0]
    # output t reads input t+i-3: frame 5 is tap i = 5-t+3
    expect = np.zeros(T)
    for t in range(T):
        i = 5 - t + 3
        if 0 <= i < 8:
            expect[t] = i + 1
    np.testing.assert_allclose(y, expect)
    assert np.flatnonzero(y).min() == 1 and np.flatnonzero(y).max() == 8   # t-4 .. t+3 around the hot frame
    ones = rced_np.conv2d_same(np.ones((1, T, 129, 1)), np.ones((8, 1, 1, 1)), np.zeros(1))[0, :, 0, 0]
    assert ones[0] == 5 and ones[T - 1] == 4 and ones[5] == 8  # t=0 sees rows 0..4 only


def test_batch_norm_known_values_and_eps():
    y = np.full((1, 1, 129, 2), 3.0)
    out = rced_np.batch_norm_inference(y, [2.0, 1.0], [0.5, -1.0], [1.0, 3.0], [4.0 - 1e-3, 0.0])
    np.testing.assert_allclose(out[0, 0, 0], [2.0 * (3 - 1) / 2.0 + 0.5, -1.0], rtol=1e-12)
    # var = 0 -> divides by sqrt(1e-3), the TF default epsilon
    out = rced_np.batch_norm_inference(np.ones((1, 1, 1, 1)), [1.0], [0.0], [0.0], [0.0])
    np.testing.assert_allclose(out.ravel(), [1 / np.sqrt(1e-3)])


def test_skip_is_added_before_relu_in_conv_bn_relu():
    x = np.ones((1, 1, 129, 1))
    k = np.full((1, 1, 1, 1), -2.0)
    skip = np.full((1, 1, 129, 1), 1.5)
    y = rced_np.conv_bn_relu(x, k, [0.0], None, skip, True)
    np.testing.assert_allclose(y, 0.0)            # relu(-2 + 1.5) = 0, not relu(-2) + 1.5
    y = rced_np.conv_bn_relu(x, k, [0.0], None, skip, False)
    np.testing.assert_allclose(y, -0.5)


def test_v3_block_skip_is_added_after_relu():
    """model.py:75-76: encode_3 + skip_input with no ReLU after -> CD outputs can be negative only via skip."""
    w = rced_np.make_weights("FullyCNNV3", seed=3)
    x = rced_np.make_input(1, 8, seed=5)
    tens = rced_np.forward("FullyCNNV3", w, x, return_all=True)
    layers = L.v3_layers()
    for i, l in enumerate(layers):
        if l.skip_post >= 0:
            pre = tens[i + 1] - tens[l.skip_post]
            assert pre.min() >= -1e-12                       # relu output
            assert np.abs(tens[l.skip_post]).max() > 0
    assert [l.skip_post for l in layers if l.skip_post >= 0] == [6, 3]   # CD1 <- CE2, CD2 <- CE1


@pytest.mark.parametrize("net_work,tag,_v", NETS)
def test_three_restatements_agree(net_work, tag, _v, built):
    import torch
    from oracle import torch_ref
    w = rced_np.make_weights(net_work, seed=7)
    x = rced_np.make_input(2, 11, seed=8)
    a = rced_np.forward(net_work, w, x)
    b = rced_c.forward(net_work, w, x, np.float64)
    c = torch_ref.TorchRef(net_work, w, torch.float64)(x).numpy()
    assert rel_err(b, a) < 1e-12
    assert rel_err(c, a) < 1e-12
    assert rel_err(rced_c.forward(net_work, w, x, np.float32), a) < 2e-5
    assert rel_err(torch_ref.TorchRef(net_work, w)(x).numpy(), a) < 2e-5


@pytest.mark.parametrize("net_work,tag,_v", NETS)
def test_oracle_matches_committed_golden(net_work, tag, _v, built):
    w, g = load_golden(tag)
    for key in ("small", "long", "c1"):
        y = rced_c.forward(net_work, w, g["x_" + key], np.float64)
        assert rel_err(y, g["y_" + key]) < 1e-12
    assert sum(v.size for k, v in w.items() if not k.endswith(("moving_mean", "moving_variance"))) == \
        L.param_count(L.layers_for(net_work))


def test_time_receptive_field_is_eight_frames():
    """Only the first conv has kh = 8 (model.py:11,37,81): frame t depends on input frames t-3..t+4."""
    for net_work in ("FullyCNN", "FullyCNNV2", "FullyCNNV3"):
        w = rced_np.make_weights(net_work, seed=1)
        x = rced_np.make_input(1, 20, seed=2)
        y = rced_np.forward(net_work, w, x)
        t = 9
        y_win = rced_np.forward(net_work, w, x[:, t - 3:t + 5])
        np.testing.assert_allclose(y_win[0, 3], y[0, t], atol=1e-12)
        x2 = x.copy()
        x2[0, t + 5] += 1.0
        x2[0, t - 4] += 1.0
        np.testing.assert_allclose(rced_np.forward(net_work, w, x2)[0, t], y[0, t], atol=1e-12)


def test_bf16_emulation_rounding_and_scale():
    """oracle.rced_np.bf16_round is round-to-nearest-even to 8 mantissa bits (checked against torch's bfloat16 cast),
    and the bf16 emulation of R-CED stays within a percent of the fp32 restatement."""
    import torch
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.standard_normal(4096).astype(np.float32) * 10.0 ** rng.integers(-6, 6, 4096),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.0e38, 1e-40], np.float32)])
    assert np.array_equal(rced_np.bf16_round(a), torch.from_numpy(a).bfloat16().float().numpy())
    assert np.array_equal(rced_np.bf16_round(rced_np.bf16_round(a)), rced_np.bf16_round(a))      # idempotent
    for net_work in ("FullyCNN", "FullyCNNV2"):
        w = rced_np.make_weights(net_work, seed=4)
        x = rced_np.make_input(1, 10, seed=5)
        y32, y16 = rced_np.forward(net_work, w, x), rced_np.forward_bf16(net_work, w, x)
        err = np.abs(y16 - y32).max() / np.abs(y32).max()
        assert 1e-4 < err < 3e-2, err          # really bf16 (not fp32), and no worse than 15 layers of 8-bit mantissas


def test_same_padding_agrees_with_a_third_partys_same_rule():
    """An independent check of the `SAME` convention the restatement hard-codes (module.py:27 passes padding='SAME';
    TF splits k-1 as (k-1)//2 before, the rest after: 3 / 4 for the 8-tall first kernels).  torch.nn.functional.conv2d's
    padding='same' implements the same rule for even kernels (the extra element goes AFTER) and, like TF, computes a
    cross-correlation: every kernel shape the three nets use must agree."""
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(3)
    for kh, kw, cin, cout in ((8, 13, 1, 12), (8, 11, 1, 10), (8, 9, 1, 18), (1, 5, 18, 30), (1, 9, 30, 8), (1, 7, 23, 25),
                              (1, 129, 8, 1), (4, 6, 2, 3)):
        x = rng.standard_normal((2, 11, 129, cin))
        k = rng.standard_normal((kh, kw, cin, cout))
        b = rng.standard_normal(cout)
        ours = rced_np.conv2d_same(x, k, b, np.float64)
        theirs = Fn.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(k).permute(3, 2, 0, 1).contiguous(),
                           torch.from_numpy(b), padding="same").permute(0, 2, 3, 1).numpy()
        assert np.abs(ours - theirs).max() < 1e-10 * np.abs(theirs).max(), (kh, kw)
        assert tuple(rced_np.same_pad(kh)) == ((kh - 1) // 2, kh - 1 - (kh - 1) // 2)


@pytest.mark.parametrize("net_work,tag,_v", NETS)
def test_matmul_restatement_agrees_with_the_numpy_one(net_work, tag, _v):
    """oracle/infer_ref.py (tap-wise torch matmuls; what the GPU tests run in float64 on the device to check EVERY frame of
    a full-size forward) against oracle/rced_np.py on shapes that cover the time halo, a batch that is cut into chunks,
    and non-trivial BatchNorm statistics."""
    from oracle import infer_ref
    w = rced_np.make_weights(net_work, seed=77)
    for n, t in ((1, 1), (3, 11), (5, 9)):
        x = rced_np.make_input(n, t, seed=31 * n + t)
        ref = rced_np.forward(net_work, w, x, np.float64)
        got = infer_ref.forward(net_work, w, x, utterances_per_chunk=2).numpy()
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 1e-12
