"""GPU parity of the training step (rced_train_step) against the fp64 autograd restatement
(oracle/train_ref.py) and the committed train-step vector (tests/golden/train_v3.npz).
Tolerances: data are fp32 with fp64 reductions; gradients are compared relative to the tensor's largest
entry.  After an Adam step every parameter moves by ~lr*sign(g), so variables are compared through the
fraction of entries that moved differently (entries whose gradient is rounding noise -- e.g. the conv bias
in front of a BatchNorm, whose true gradient is exactly zero -- may legitimately take the other sign)."""

import os

import numpy as np
import pytest

from conftest import NETS, ROOT
from oracle import rced_np, train_ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


# Gradients of the LAST layers are compared tightly.  Deeper ones go through ReLU masks recomputed in
# fp32: a pre-activation within rounding of zero takes the other branch than in the fp64 oracle, which moves
# a sum of ~8k terms by one term (~1e-2 relative) and propagates at ~1e-3 to everything upstream.  The
# oracle itself shows the same: its float32 run differs from its float64 run by up to 1.5e-3.
TIGHT, LOOSE, COS = 1e-4, 3e-2, 0.9999


def moved_differently(v_gpu, v_ref, v0, lr):
    d = np.abs((np.asarray(v_gpu, np.float64) - v0) - (np.asarray(v_ref, np.float64) - v0))
    return float((d > 0.1 * lr).mean())


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(ROOT, "tests", "golden", "train_v3.npz"))
    return {k: z[k] for k in z.files}


def test_two_steps_match_committed_vector(gold, built):
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV3", seed=42)
    tr = FullyCNNTrainer("FullyCNNV3", batch_size=4, lr=1e-3, warmup_steps=4000.0, weights=w)
    loss1, _, step1 = tr.fit_step(gold["x"], gold["y"])
    assert step1 == 1 and abs(loss1 - gold["loss"][0]) <= 1e-5 * gold["loss"][0]
    g = tr.gradients()
    for k in ("decode_final/kernel", "decode_final/bias"):
        assert rel(g[k], gold["g1:" + k]) < TIGHT, k
    for k in ("CE1_encode_1/kernel", "CE1_encode_1/batch_norm/gamma"):
        assert rel(g[k], gold["g1:" + k]) < LOOSE and cosine(g[k], gold["g1:" + k]) > COS, k
    v1 = tr.variables()
    assert abs(tr.lr - gold["lr"][1]) <= 1e-12            # Noam schedule for the next step (trainer.py:215)
    loss2, _, step2 = tr.fit_step(gold["x"], gold["y"])
    assert step2 == 2 and abs(loss2 - gold["loss"][1]) <= 2e-3 * gold["loss"][1]
    v2 = tr.variables()
    for name, arr in v1.items():
        ref1, ref2, v0 = gold["v1:" + name], gold["v2:" + name], np.asarray(w[name], np.float64)
        if "moving_" in name:
            # step 2 sees conv biases that moved by +-lr in step 1 on rounding-noise gradients (see below),
            # which shifts that layer's batch mean by up to lr: 0.01 * 1e-3 on the moving mean
            assert rel(arr, ref1) < 1e-5 and rel(v2[name], ref2) < 5e-4, name
        elif name.endswith("/bias") and name != "decode_final/bias":
            continue       # bias in front of BatchNorm: true gradient is 0, its Adam move is rounding noise
        else:
            # at most one entry (or 2 %) may have a gradient small enough for its sign to be rounding noise
            assert moved_differently(arr, ref1, v0, 1e-3) <= max(0.02, 1.01 / arr.size), name


def net_layers_last(net_work):
    return {"FullyCNN": "decode_5", "FullyCNNV2": "decode_8", "FullyCNNV3": "decode_final"}[net_work]


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_loss_and_gradients_all_nets(net_work, tag, variant, built):
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights(net_work, seed=7)
    x = rced_np.make_input(3, 9, seed=70 + variant)
    y = rced_np.make_input(3, 9, seed=80 + variant)
    ref = train_ref.TrainRef(net_work, w, batch_size=8)     # configured batch size != dynamic N on purpose
    loss_ref, grads_ref, _ = ref.loss_and_grads(x, y)
    tr = FullyCNNTrainer(net_work, batch_size=8, lr=1e-4, weights=w)
    loss, _, step = tr.train_step(x, y)
    assert step == 1 and abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    g = tr.gradients()
    for name, gr in grads_ref.items():
        if name.endswith("/bias") and (name[:-5] + "/batch_norm/gamma") in grads_ref:
            assert np.abs(g[name]).max() <= 1e-3 * max(np.abs(g[name[:-5] + "/kernel"]).max(), 1.0)   # ~0
            continue
        last = name.startswith(net_layers_last(net_work))
        assert rel(g[name], gr.numpy()) < (TIGHT if last else LOOSE), name
        assert cosine(g[name], gr.numpy()) > COS, name


def test_training_reduces_the_loss_and_moves_bn_statistics(built):
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV2", seed=3)
    x = rced_np.make_input(4, 12, seed=1)
    y = 0.5 * x
    tr = FullyCNNTrainer("FullyCNNV2", batch_size=4, lr=2e-3, warmup_steps=1.0, weights=w)
    losses = [tr.train_step(x, y)[0] for _ in range(8)]
    assert losses[-1] < 0.7 * losses[0]
    v = tr.variables()
    assert np.abs(v["encode_1/batch_norm/moving_mean"] - w["encode_1/batch_norm/moving_mean"]).max() > 1e-4
    assert tr.global_step == 8


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_mfma_and_generic_training_kernels_agree(net_work, tag, variant, built, monkeypatch):
    """RCED_TRAIN_MFMA=0 keeps the direct-convolution kernels for every layer; the MFMA kernels
    (kernels_train_mfma.h) must give the same step.  Ragged shape: 5 x 7 = 35 frames (odd: half-empty tile)."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights(net_work, seed=17)
    x = rced_np.make_input(5, 7, seed=31)
    y = rced_np.make_input(5, 7, seed=32)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RCED_TRAIN_MFMA", mode)
        tr = FullyCNNTrainer(net_work, batch_size=5, lr=1e-3, weights=w)
        loss, _, _ = tr.train_step(x, y)
        out[mode] = (loss, tr.gradients(), tr.variables())
        tr.close()
    assert abs(out["0"][0] - out["1"][0]) <= 1e-6 * abs(out["0"][0])
    last = net_layers_last(net_work)
    for name, g0 in out["0"][1].items():
        g1 = out["1"][1][name]
        if "moving_" in name or (name.endswith("/bias") and not name.startswith(last)):
            continue                                   # not trainable / rounding noise in front of BatchNorm (see above)
        # one ReLU-mask flip (a pre-activation within fp32 rounding of zero lands on the other side in the two
        # kernels) moves a single entry of a per-channel sum by one term: bound the entry at 2 x LOOSE, the
        # direction of the whole gradient tightly
        assert rel(g1, g0) < (TIGHT if name.startswith(last) else 2 * LOOSE), name
        assert cosine(g1, g0) > COS, name
    for name in out["0"][2]:
        if "moving_" in name:
            assert rel(out["1"][2][name], out["0"][2][name]) < 1e-5, name


def test_checkpoint_resume_continues_the_run(built, tmp_path):
    """trainer.py:50-65 continue_train: save variables + Adam slots + global_step as a TF V2 checkpoint, resume in a
    new trainer, and the continued run follows the uninterrupted one bit for bit (weight gradients are reduced in a
    fixed order; the checkpoint stores fp32 values exactly)."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer, tf_checkpoint
    w = rced_np.make_weights("FullyCNNV3", seed=21)
    x, y = rced_np.make_input(4, 12, seed=22), 0.5 * rced_np.make_input(4, 12, seed=23)
    kw = dict(batch_size=4, lr=1e-3, warmup_steps=50.0)
    a = FullyCNNTrainer("FullyCNNV3", weights=w, **kw)
    for _ in range(3):
        a.fit_step(x, y)
    prefix = a.save_checkpoint(str(tmp_path / "RCED_FullyCNNV3_0_2"))
    stored = tf_checkpoint.read_checkpoint(prefix)
    assert int(stored["global_step"]) == 3 and "decode_final/kernel/Adam_1" in stored
    assert "CE1_encode_1/batch_norm/moving_mean/Adam" not in stored                # slots only for trainable variables
    assert abs(float(stored["beta1_power"]) - 0.9 ** 4) < 1e-7
    b = FullyCNNTrainer.from_checkpoint(prefix, "FullyCNNV3", **kw)
    assert b.global_step == 3 and abs(b.lr - a.lr) <= 1e-12 * a.lr
    la = [a.fit_step(x, y)[0] for _ in range(2)]
    lb = [b.fit_step(x, y)[0] for _ in range(2)]
    assert a.global_step == b.global_step == 5
    assert la == lb
    va, vb = a.variables(), b.variables()
    for name in va:
        assert np.array_equal(va[name], vb[name]), name
    # without the optimizer state the checkpoint still serves the reference's test / infer graphs
    p2 = a.save_checkpoint(str(tmp_path / "weights_only"), with_optimizer=False)
    assert "decode_final/kernel/Adam" not in tf_checkpoint.read_checkpoint(p2)


@pytest.mark.parametrize("net_work", ["FullyCNNV3", "FullyCNN", "FullyCNNV2"])
def test_mfma_kernels_agree_with_generic_on_a_multi_tile_ragged_batch(net_work, built, monkeypatch):
    """4847 frames (odd) = 2424 two-frame tiles: more tiles than persistent workgroups, so every MFMA training kernel
    runs its prefetch-next-tile loop and ends on a half-empty tile; RCED_TRAIN_MFMA=0 is the direct-conv reference.
    All three nets; R-CED V2's odd channel counts run even-padded inside the MFMA trainer (phantom channels), so for
    it this also checks that the padded layout trains exactly the reference's variables."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights(net_work, seed=27)
    x = rced_np.make_input(37, 131, seed=41)
    y = rced_np.make_input(37, 131, seed=42)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RCED_TRAIN_MFMA", mode)
        tr = FullyCNNTrainer(net_work, batch_size=37, lr=1e-3, weights=w)
        loss, _, _ = tr.train_step(x, y)
        out[mode] = (loss, tr.gradients())
        tr.close()
    assert abs(out["0"][0] - out["1"][0]) <= 1e-6 * abs(out["0"][0])
    for name, g0 in out["0"][1].items():
        last = name.startswith("decode_final") or name.startswith({"FullyCNN": "decode_5", "FullyCNNV2": "decode_8"}.get(net_work, "decode_final"))
        if "moving_" in name or (name.endswith("/bias") and not last):
            continue
        g1 = out["1"][1][name]
        assert rel(g1, g0) < (TIGHT if last else 2 * LOOSE), name
        assert cosine(g1, g0) > COS, name


@pytest.mark.parametrize("switch", ["RCED_TRAIN_FUSE_SUMS", "RCED_TRAIN_FUSE_BWD"])
def test_sums_from_the_dgrad_agree_with_the_separate_pass(switch, built, monkeypatch):
    """RCED_TRAIN_FUSE_BWD: wgrad + dgrad of the 18->30 and 30->8 layers in one kernel (tmm::bwd_fused_mfma, sums from the
    transformed x tile) against the separate wgrad / dgrad kernels (sums from the dgrad's z tile).  RCED_TRAIN_FUSE_SUMS:
    CR-CED's plain layers (18 and 30 channels) get their BatchNorm-backward sums S1, S2 from the epilogue of the dgrad
    that writes their gradient (tmm::SumArgs: z tile by LDS-DMA, masked sums, sums_fix); RCED_TRAIN_FUSE_SUMS=0 is the
    bwd_route2 pass over g and z they replace.  Same ragged multi-tile batch as above: the partial last tile takes the
    ordinary-load path of ztile_fetch.  The two differ only in summation order (fp32 per-tile partials vs fp64 per element)."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV3", seed=27)
    x = rced_np.make_input(37, 131, seed=41)
    y = rced_np.make_input(37, 131, seed=42)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv(switch, mode)
        tr = FullyCNNTrainer("FullyCNNV3", batch_size=37, lr=1e-3, weights=w)
        loss, _, _ = tr.train_step(x, y)
        out[mode] = (loss, tr.gradients())
        tr.close()
    assert out["0"][0] == out["1"][0]                       # the forward is the same code
    for name, g0 in out["0"][1].items():
        if "moving_" in name or (name.endswith("/bias") and not name.startswith("decode_final")):
            continue                                        # a bias in front of BatchNorm has gradient 0: rounding noise only
        assert rel(out["1"][1][name], g0) < 1e-4, name
        assert cosine(out["1"][1][name], g0) > 1 - 1e-8, name


def test_bf16_pipe_kernels_agree_with_the_fp32_mfma_kernels(built, monkeypatch):
    """RCED_TRAIN_X6: the 18 -> 30 forward convolutions (tmm::conv_x6_fwd) and the output layer's forward and dgrad
    (kernels_final_x6.h) compute fp32 products as six bf16 MFMAs over three-part operands; RCED_TRAIN_X6=0 keeps the fp32 MFMA
    kernels.  Same ragged multi-tile batch as above.  The loss must agree to 1e-6.  The gradients are held to the bound the
    fp64 comparison of this batch uses (5e-3 of the tensor's largest entry, cosine 1 - 2.5e-5): forward values that differ in
    the last bits move a few of 625 k ReLU masks, and a deep layer's gradient changes by whole terms -- two fp32 arithmetics
    differ from each other as much as each differs from fp64 (measured: 2.3e-3 on CE1_encode_1/kernel; against fp64 2.7e-3 /
    4.0e-3).  That the three-part products themselves are exact to fp32 rounding is what the tight element-wise test (inputs
    without ReLU ties, 5e-6) and tools/micro/f32_on_bf16.hip establish."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV3", seed=27)
    x = rced_np.make_input(37, 131, seed=41)
    y = rced_np.make_input(37, 131, seed=42)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RCED_TRAIN_X6", mode)
        tr = FullyCNNTrainer("FullyCNNV3", batch_size=37, lr=1e-3, weights=w)
        pred = np.array(tr.valid_step(x))           # the train-mode forward alone: every tile of both arithmetics, element-wise
        loss, _, _ = tr.train_step(x, y)
        out[mode] = (loss, tr.gradients(), pred)
        tr.close()
    assert abs(out["0"][0] - out["1"][0]) <= 1e-6 * abs(out["0"][0])
    fwd = np.abs(out["0"][2] - out["1"][2]).max() / np.abs(out["0"][2]).max()
    assert fwd < 2e-5, fwd                          # (measured ~1e-6: last-bit differences carried through sixteen layers)
    worst = 0.0
    for name, g0 in out["0"][1].items():
        if "moving_" in name or (name.endswith("/bias") and not name.startswith("decode_final")):
            continue                                        # a bias in front of BatchNorm has gradient 0: rounding noise only
        worst = max(worst, rel(out["1"][1][name], g0))
        assert rel(out["1"][1][name], g0) < 5e-3, name
        assert cosine(out["1"][1][name], g0) > 1 - 2.5e-5, name
    print("\n[bf16-pipe kernels vs fp32 MFMA kernels] loss %.3e apart, forward %.2e, worst gradient %.2e of its tensor's max"
          % (abs(out["0"][0] - out["1"][0]) / abs(out["0"][0]), fwd, worst))


@pytest.mark.parametrize("switch", ["RCED_TRAIN_FUSE_ACT", "RCED_TRAIN_FUSE_DZ"])
def test_fused_staging_switches_agree_with_the_materialised_tensors(switch, built, monkeypatch):
    """RCED_TRAIN_FUSE_ACT=0 materialises every activation tensor (relu(bn(z)) written by bn_act_fwd2 and read back) instead of
    rebuilding it in the consumers' staging; RCED_TRAIN_FUSE_DZ=0 applies the BatchNorm backward in place instead of inside
    the wgrad / dgrad staging.  Both are the fallbacks the fused forms replaced, and the only callers of the kernels' plain
    variants (among them tmm::conv_x6_fwd without a transform): same ragged multi-tile batch, same step up to rounding -- the
    loss to 1e-6, the train-mode forward element-wise (the arithmetic is the same: measured identical), every gradient to 1e-4
    of its tensor's largest entry (summation order)."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV3", seed=27)
    x = rced_np.make_input(37, 131, seed=41)
    y = rced_np.make_input(37, 131, seed=42)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv(switch, mode)
        tr = FullyCNNTrainer("FullyCNNV3", batch_size=37, lr=1e-3, weights=w)
        pred = np.array(tr.valid_step(x))
        loss, _, _ = tr.train_step(x, y)
        out[mode] = (loss, tr.gradients(), pred)
        tr.close()
    assert abs(out["0"][0] - out["1"][0]) <= 1e-6 * abs(out["0"][0])
    fwd = np.abs(out["0"][2] - out["1"][2]).max() / np.abs(out["0"][2]).max()
    assert fwd < 1e-6, fwd                          # (measured 0: the same arithmetic either way)
    worst = 0.0
    for name, g0 in out["0"][1].items():
        if "moving_" in name or (name.endswith("/bias") and not name.startswith("decode_final")):
            continue
        worst = max(worst, rel(out["1"][1][name], g0))
        assert rel(out["1"][1][name], g0) < 1e-4, name   # (measured 2.5e-6 / 2.8e-6: summation order only -- the forward is identical)
        assert cosine(out["1"][1][name], g0) > 1 - 1e-8, name
    print("\n[%s = 0 vs 1] loss %.3e apart, forward %.2e, worst gradient %.2e of its tensor's max"
          % (switch, abs(out["0"][0] - out["1"][0]) / abs(out["0"][0]), fwd, worst))


def test_padded_layout_round_trips_variables_and_adam_state(built):
    """R-CED V2 trains in an even-padded internal layout; what crosses the ABI (variables, gradients, Adam slots) is the
    reference's unpadded layout: get -> set -> get is the identity and a resumed trainer continues like the original."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV2", seed=31)
    x, y = rced_np.make_input(3, 10, seed=32), 0.5 * rced_np.make_input(3, 10, seed=33)
    a = FullyCNNTrainer("FullyCNNV2", batch_size=3, lr=1e-3, weights=w)
    v0 = a.variables()
    for name, ref in w.items():
        assert v0[name].shape == np.asarray(ref).shape and np.array_equal(v0[name], np.asarray(ref, np.float32)), name
    a.train_step(x, y)
    a.train_step(x, y)
    m, v, step = a.optimizer_state()
    b = FullyCNNTrainer("FullyCNNV2", batch_size=3, lr=1e-3, weights=a.variables())
    b.load_optimizer_state(m, v, step)
    m2, v2, step2 = b.optimizer_state()
    assert step2 == step == 2
    for d, d2 in ((m, m2), (v, v2)):
        for name in d:
            assert np.array_equal(d[name], d2[name]), name
    la, lb = a.train_step(x, y)[0], b.train_step(x, y)[0]
    assert la == lb
    a.close()
    b.close()


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_valid_step_is_the_training_graph_forward_and_changes_nothing(net_work, tag, variant, built):
    """trainer.py:245-250 fetches `pred` of the is_training=True graph: batch-statistics BatchNorm, no updates."""
    import torch
    from fullycnnspeechenhancement_amd import FullyCNNTrainer, build_model
    from oracle import train_ref
    w = rced_np.make_weights(net_work, seed=50 + variant)
    x = rced_np.make_input(3, 11, seed=51)
    tr = FullyCNNTrainer(net_work, batch_size=3, lr=1e-3, weights=w)
    before = tr.variables()
    pred = tr.valid_step(x)
    ref = train_ref.TrainRef(net_work, w, 3).forward_train(torch.from_numpy(x).double())[0].detach().numpy()
    assert pred.shape == x.shape and pred.dtype == np.float32
    assert rel(pred, ref) < 1e-4
    after = tr.variables()
    for name in before:                                   # moving statistics included: nothing was updated
        assert np.array_equal(before[name], after[name]), name
    assert tr.global_step == 0
    # it is NOT the inference graph (moving statistics): the two differ for a freshly initialised net
    inf = build_model(net_work, False, weights=w)(x)
    assert rel(pred, inf) > 1e-3
    on_dev = tr.valid_step(torch.from_numpy(x).cuda())
    assert on_dev.is_cuda and np.array_equal(on_dev.cpu().numpy(), pred)
    tr.close()


def screened_input(ref, n, t, seed0, margin=1e-5):
    """Reject-and-redraw: the first seeded input for which NO value entering a ReLU is within `margin` of zero in the
    fp64 restatement (the GPU's fp32 pre-activations differ from it by ~1e-6), so that no ReLU mask can flip and
    gradients can be compared tightly.  ~1e5 pre-activations of unit scale: about one draw in five passes."""
    for s in range(400):
        x = rced_np.make_input(n, t, seed=seed0 + s)
        if ref.min_abs_preactivation(x) > margin:
            return x, seed0 + s
    raise AssertionError("no screened input in 400 draws")


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_all_gradients_tight_on_inputs_without_relu_ties(net_work, tag, variant, built, capsys):
    """Every gradient of every layer, element-wise, within 1e-4 of the tensor's largest entry against the fp64 autograd
    restatement (trainer.py:146-147 loss, module.py:27-33 layers), on an input screened so that no ReLU argument lies
    within 1e-5 of zero -- the loose bounds of test_loss_and_gradients_all_nets exist only because of such ties."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights(net_work, seed=11)
    ref = train_ref.TrainRef(net_work, w, batch_size=4)
    x, seed = screened_input(ref, 2, 3, 5000 + 100 * variant)
    y = rced_np.make_input(2, 3, seed=77 + variant)
    loss_ref, grads_ref, _ = ref.loss_and_grads(x, y)
    tr = FullyCNNTrainer(net_work, batch_size=4, lr=1e-4, weights=w)
    loss, _, _ = tr.train_step(x, y)
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    g = tr.gradients()
    worst = {}
    for name, gr in grads_ref.items():
        gr = gr.numpy()
        if name.endswith("/bias") and (name[:-5] + "/batch_norm/gamma") in grads_ref:
            # a bias in front of BatchNorm: the true gradient is exactly 0 (the batch mean removes it)
            assert np.abs(g[name]).max() <= 1e-4 * max(np.abs(g[name[:-5] + "/kernel"]).max(), 1e-30), name
            continue
        scale = np.abs(gr).max()
        err = np.abs(g[name].astype(np.float64) - gr).max() / scale
        worst[name] = err
        assert err < TIGHT, (name, err)
    with capsys.disabled():
        k = max(worst, key=worst.get)
        print("\n[train parity] %s seed %d: loss rel err %.2e, worst gradient %s %.2e of its max (%d tensors)" % (
            net_work, seed, abs(loss - loss_ref) / abs(loss_ref), k, worst[k], len(worst)))
    tr.close()


def test_gamma_zero_and_tiny_gamma_get_their_true_gradients(built, capsys):
    """A BatchNorm channel with gamma = 0 (and beta > 0) still has d gamma = sum dy * zhat != 0 (module.py:29,
    tf.layers.batch_normalization).  The fused backward kernel recovers zhat from the transformed activations by dividing
    by gamma * rstd, which cannot work there: the step must notice and recompute that layer's sums from (g, z).  Channels
    of an 18- and a 30-channel layer (the tensors whose sums come from the fused kernel) with gamma = 0 and 2e-4."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = {k: v.copy() for k, v in rced_np.make_weights("FullyCNNV3", seed=11).items()}
    edits = (("CE2_encode_1", 5, 0.0, 0.3), ("CD1_encode_2", 17, 0.0, 0.2), ("CE3_encode_2", 3, 2e-4, 0.25),
             ("CD2_encode_1", 11, -3e-4, 0.4))
    for scope, c, gam, bet in edits:
        w[scope + "/batch_norm/gamma"][c] = gam
        w[scope + "/batch_norm/beta"][c] = bet
    ref = train_ref.TrainRef("FullyCNNV3", w, batch_size=4)
    x, seed = screened_input(ref, 2, 3, 9100)
    y = rced_np.make_input(2, 3, seed=78)
    loss_ref, grads_ref, _ = ref.loss_and_grads(x, y)
    tr = FullyCNNTrainer("FullyCNNV3", batch_size=4, lr=1e-4, weights=w)
    loss, _, _ = tr.train_step(x, y)
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    g = tr.gradients()
    worst = 0.0
    for name, gr in grads_ref.items():
        gr = gr.numpy()
        if name.endswith("/bias") and (name[:-5] + "/batch_norm/gamma") in grads_ref:
            continue
        err = np.abs(g[name].astype(np.float64) - gr).max() / np.abs(gr).max()
        worst = max(worst, err)
        assert err < TIGHT, (name, err)
    for scope, c, gam, bet in edits:      # the edited channels themselves: gamma's gradient is there and right
        gr = grads_ref[scope + "/batch_norm/gamma"].numpy()
        assert abs(gr[c]) > 1e-3 * np.abs(gr).max(), (scope, "the case is vacuous: d gamma ~ 0")
        assert abs(g[scope + "/batch_norm/gamma"][c] - gr[c]) <= TIGHT * np.abs(gr).max(), (scope, c)
    with capsys.disabled():
        print("\n[train parity] gamma = 0 / tiny gamma channels: worst gradient error %.2e of its tensor's max" % worst)
    tr.close()


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_model_is_training_true_is_the_training_graph(net_work, tag, variant, built):
    """trainer.py:165-172 builds `Model(is_training=True)`; model(x) then normalises with the statistics of the batch
    (module.py:29 with training=True) and changes no variable.  Checked against the fp64 restatement; the trainer runs
    its steps on the same library handle (`trainer.model`)."""
    import torch
    from fullycnnspeechenhancement_amd import FullyCNNTrainer, build_model
    w = rced_np.make_weights(net_work, seed=60 + variant)
    x = rced_np.make_input(3, 11, seed=61)
    ref = train_ref.TrainRef(net_work, w, 3).forward_train(torch.from_numpy(x).double())[0].detach().numpy()
    m = build_model(net_work, True, weights=w)
    assert m.is_training
    pred = m(x)
    assert isinstance(pred, np.ndarray) and pred.dtype == np.float32 and rel(pred, ref) < 1e-4
    xd = torch.from_numpy(x).cuda()
    out = torch.empty_like(xd)
    assert m(xd, out=out) is out and np.array_equal(out.cpu().numpy(), pred)
    with pytest.raises(RuntimeError):
        m.set_option("bf16", 1)                      # inference-kernel options do not exist on the training graph
    m.close()
    tr = FullyCNNTrainer(net_work, batch_size=3, lr=1e-3, weights=w)
    assert tr.model.is_training and np.array_equal(tr.model(x), pred)        # same graph, same handle as train_step
    tr.train_step(x, x)
    assert rel(tr.model(x), pred) > 1e-6                                     # the step moved the shared variables
    tr.close()


def test_full_size_config5_against_fp64_restatement_on_the_gpu(built, capsys):
    """BASELINE configs[4] at its full size (CR-CED V3, batch 256 x 512 frames; trainer.py:181-192).  Batch statistics
    couple all 16.9 M pixels, so nothing can be sampled: the whole forward is restated in float64 with plain torch
    matmuls on the same GPU (oracle/train_ref.py, conv="taps").  Checked: the loss (1e-5), every layer's batch mean and
    variance as they reach the moving statistics (momentum 0.99, unbiased variance), and the output layer's gradients
    (autograd through the fp64 restatement's last layer)."""
    import torch
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    B, T = 256, 512
    w = rced_np.make_weights("FullyCNNV3", seed=42)
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn((B, T, 129, 1), generator=g, device="cuda").abs_()
    y = 0.5 * torch.randn((B, T, 129, 1), generator=g, device="cuda").abs_()
    tr = FullyCNNTrainer("FullyCNNV3", batch_size=B, lr=1e-3, weights=w)
    loss, _, step = tr.train_step(x, y)
    assert step == 1 and np.isfinite(loss)
    grads, after = tr.gradients(), tr.variables()
    tr.close()
    torch.cuda.empty_cache()
    ref = train_ref.TrainRef("FullyCNNV3", w, B, device="cuda", conv="taps")
    with torch.no_grad():
        pred, stats = ref.forward_train(x)
        hidden = ref.last_hidden
    loss_ref = float(((y.double() - pred) ** 2).sum() / B)
    assert abs(loss - loss_ref) <= 1e-5 * loss_ref, (loss, loss_ref)
    worst = 0.0
    for scope, mean, var, n in stats:
        p = scope + "/batch_norm/"
        mm = 0.99 * np.asarray(w[p + "moving_mean"], np.float64) + 0.01 * mean.cpu().numpy()
        mv = 0.99 * np.asarray(w[p + "moving_variance"], np.float64) + 0.01 * var.cpu().numpy() * n / (n - 1)
        e1 = np.abs(after[p + "moving_mean"] - mm).max() / max(np.abs(mm).max(), 1e-30)
        e2 = np.abs(after[p + "moving_variance"] - mv).max() / np.abs(mv).max()
        worst = max(worst, e1, e2)
        assert e1 < 1e-5 and e2 < 1e-5, (scope, e1, e2)
    # output layer: d loss / d (kernel, bias) by autograd through the restatement's last layer only
    kf = ref.vars["decode_final/kernel"].detach().clone().requires_grad_(True)
    bf = ref.vars["decode_final/bias"].detach().clone().requires_grad_(True)
    hp = torch.nn.functional.pad(hidden, (0, 0, 64, 64))
    out = sum(hp[:, :, j:j + 129, :] @ kf[0, j] for j in range(129)) + bf
    (((y.double() - out) ** 2).sum() / B).backward()
    eg = {}
    for name, gr in (("decode_final/kernel", kf.grad), ("decode_final/bias", bf.grad)):
        gr = gr.cpu().numpy()
        eg[name] = np.abs(grads[name] - gr).max() / np.abs(gr).max()
        assert eg[name] < TIGHT, (name, eg[name])
    del ref, pred, hidden, hp, out
    torch.cuda.empty_cache()
    with capsys.disabled():
        print("\n[config 5 full size] loss %.6f vs fp64 %.6f (rel %.1e); BN statistics worst rel err %.1e; "
              "decode_final grads %.1e / %.1e" % (loss, loss_ref, abs(loss - loss_ref) / loss_ref, worst,
                                                 eg["decode_final/kernel"], eg["decode_final/bias"]))


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_training_steps_are_bit_reproducible(net_work, tag, variant, built):
    """Weight gradients are summed from per-wave slices in a fixed order (tmm::wg_reduce), not with fp32 atomics: two
    trainers fed the same batches produce the same bits -- losses, every gradient, every variable after three steps.
    (The reference on one device is deterministic per op order; round 1's atomics were not.)  Multi-tile ragged batch:
    more tiles than persistent workgroups."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights(net_work, seed=91)
    x = rced_np.make_input(37, 131, seed=92)
    y = rced_np.make_input(37, 131, seed=93)
    runs = []
    for _ in range(2):
        tr = FullyCNNTrainer(net_work, batch_size=37, lr=1e-3, warmup_steps=10.0, weights=w)
        losses = [tr.fit_step(x, y)[0] for _ in range(3)]
        runs.append((losses, tr.gradients(), tr.variables()))
        tr.close()
    assert runs[0][0] == runs[1][0]
    for name in runs[0][1]:
        assert np.array_equal(runs[0][1][name], runs[1][1][name]), "gradient " + name
    for name in runs[0][2]:
        assert np.array_equal(runs[0][2][name], runs[1][2][name]), "variable " + name


def test_atomic_wgrad_option_agrees_with_the_ordered_reduction(built, monkeypatch):
    """RCED_TRAIN_DET=0 keeps round 1's fp32 atomics: same gradients up to summation order."""
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights("FullyCNNV3", seed=94)
    x, y = rced_np.make_input(9, 40, seed=95), rced_np.make_input(9, 40, seed=96)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RCED_TRAIN_DET", mode)
        tr = FullyCNNTrainer("FullyCNNV3", batch_size=9, lr=1e-3, weights=w)
        tr.train_step(x, y)
        out[mode] = tr.gradients()
        tr.close()
    for name, g0 in out["0"].items():
        if "moving_" in name or (name.endswith("/bias") and not name.startswith("decode_final")):
            continue
        assert rel(out["1"][name], g0) < 1e-5, name


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_every_gradient_on_the_multi_tile_ragged_batch_against_fp64_autograd(net_work, tag, variant, built, capsys, monkeypatch):
    """The multi-tile bar.  4847 frames (37 x 131, odd) = 2424 two-frame tiles: every persistent training kernel walks its
    prefetch-next-tile loop, ends on a half-empty tile, and the weight gradients go through the per-wave slices and
    wg_reduce.  EVERY gradient of every layer against the fp64 autograd restatement run on the same GPU with plain torch
    matmuls (oracle/train_ref.py, conv="taps"; trainer.py:146-147 loss, module.py:27-33 layers) -- not against this
    repo's own direct-convolution kernels.
    Bound: 5e-3 of the tensor's largest entry.  What is measured is fp32 summation noise, not kernel error: a gradient
    entry is a sum over 625 k pixels of terms of either sign, so an fp32 accumulation that is exact to ~1e-7 of the sum
    of the terms' magnitudes is off by ~1e-7 x sqrt(N) ~ 1e-4 ... 1e-3 of the result (measured worst entries: R-CED V1
    7e-4, V2 2.3e-3, CR-CED 2.8e-3, medians ~1e-4; on 6-frame inputs the same kernels are at 1e-6 ... 5e-6, see
    test_all_gradients_tight_on_inputs_without_relu_ties).  The direct-convolution path (RCED_TRAIN_MFMA=0: fp32 FMA
    chains in another order) is run beside it and printed as the noise yardstick; a wrong border tap or tile seam moves
    whole terms (>= 1e-2)."""
    import torch
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    w = rced_np.make_weights(net_work, seed=27)
    x = rced_np.make_input(37, 131, seed=41)
    y = rced_np.make_input(37, 131, seed=42)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RCED_TRAIN_MFMA", mode)
        tr = FullyCNNTrainer(net_work, batch_size=37, lr=1e-3, weights=w)
        loss, _, _ = tr.train_step(x, y)
        got[mode] = (loss, tr.gradients())
        tr.close()
    loss, g = got["1"]
    torch.cuda.empty_cache()
    ref = train_ref.TrainRef(net_work, w, batch_size=37, device="cuda", conv="taps")
    loss_ref, grads_ref, _ = ref.loss_and_grads(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref), (loss, loss_ref)
    worst, yard = {}, {}
    for name, gr in grads_ref.items():
        gr = gr.cpu().numpy()
        if name.endswith("/bias") and (name[:-5] + "/batch_norm/gamma") in grads_ref:
            # a bias in front of BatchNorm: the true gradient is exactly 0 (the batch mean removes it)
            assert np.abs(g[name]).max() <= 1e-3 * max(np.abs(g[name[:-5] + "/kernel"]).max(), 1e-30), name
            continue
        worst[name] = np.abs(g[name].astype(np.float64) - gr).max() / np.abs(gr).max()
        yard[name] = np.abs(got["0"][1][name].astype(np.float64) - gr).max() / np.abs(gr).max()
        assert worst[name] < 5e-3, (name, worst[name], yard[name])
        assert cosine(g[name], gr) > 1 - 2.5e-5, name     # |error| <= 5e-3 of the max bounds 1 - cos at ~1.25e-5
    del ref, grads_ref
    torch.cuda.empty_cache()
    with capsys.disabled():
        k = max(worst, key=worst.get)
        print("\n[train parity, 37 x 131] %s: loss rel err %.2e; worst gradient %s %.2e of its max, median %.2e (%d tensors); "
              "direct-convolution path: worst %.2e, median %.2e" % (
                  net_work, abs(loss - loss_ref) / abs(loss_ref), k, worst[k], float(np.median(list(worst.values()))), len(worst),
                  max(yard.values()), float(np.median(list(yard.values())))))


def test_full_size_config5_mid_network_bn_gradients(built, capsys):
    """BASELINE configs[4] at full size, a DEEP layer: d beta and d gamma of CD1_encode_2 (18 -> 30 channels, block 4) --
    they are the BatchNorm-backward sums S1, S2 the fused backward kernel forms from its transformed x tile, after the
    gradient has come back through CD2's three layers and CD1_decode (prefetch-next-tile loops, bwd_fused_mfma,
    wg_reduce at 128 tiles per workgroup).  The fp64 restatement runs the net up to that layer without autograd and
    records only the five layers behind it (oracle/train_ref.py mid_layer_bn_grads)."""
    import torch
    from fullycnnspeechenhancement_amd import FullyCNNTrainer
    import bench
    B, T = 256, 512
    w = rced_np.make_weights("FullyCNNV3", seed=42)
    x = torch.from_numpy(bench.synthetic_magnitudes((B, T, 129, 1), 1234)).cuda()
    y = torch.from_numpy(bench.synthetic_magnitudes((B, T, 129, 1), 1235)).cuda()
    tr = FullyCNNTrainer("FullyCNNV3", batch_size=B, lr=1e-3, weights=w)
    loss, _, _ = tr.train_step(x, y)
    grads = tr.gradients()
    tr.close()
    torch.cuda.empty_cache()
    ref = train_ref.TrainRef("FullyCNNV3", w, B, device="cuda", conv="taps")
    dbeta, dgamma, loss_ref = ref.mid_layer_bn_grads(x, y, "CD1_encode_2")
    assert abs(loss - loss_ref) <= 1e-5 * loss_ref
    e = {}
    for name, gr in (("CD1_encode_2/batch_norm/beta", dbeta), ("CD1_encode_2/batch_norm/gamma", dgamma)):
        gr = gr.cpu().numpy()
        e[name] = np.abs(grads[name] - gr).max() / np.abs(gr).max()
        assert e[name] < 2e-3, (name, e[name])
    del ref
    torch.cuda.empty_cache()
    with capsys.disabled():
        print("\n[config 5 full size, CD1_encode_2] d beta %.1e, d gamma %.1e of their max" % (
            e["CD1_encode_2/batch_norm/beta"], e["CD1_encode_2/batch_norm/gamma"]))
