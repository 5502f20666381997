"""TF-free checkpoint / frozen-graph readers.  No TF-written file exists in this image, so
these tests pin the codecs to published known answers (CRC-32C, snappy, table layout) and to round trips."""
import os
import struct

import numpy as np
import pytest

from fullycnnspeechenhancement_amd import spec, tf_checkpoint as tfc, weights as W
from oracle import rced_np


def test_crc32c_known_answers():
    # RFC 3720 B.4 / LevelDB crc32c_test.cc
    assert tfc.crc32c(b"123456789") == 0xE3069283
    assert tfc.crc32c(bytes(32)) == 0x8A9136AA
    assert tfc.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfc.crc32c(b"hello world") == tfc.crc32c(b" world", tfc.crc32c(b"hello"))       # extendable
    # LevelDB's mask is a rotation + constant and must not be the identity
    c = tfc.crc32c(b"foo")
    assert tfc._mask_crc(c) != c and tfc._mask_crc(tfc._mask_crc(c)) != c


def test_snappy_decoder_literal_and_overlapping_copy():
    # length 8 | literal "ab" | copy(offset 2, length 6)  ->  "abababab"
    assert tfc._snappy_decompress(bytes([8, 0x04]) + b"ab" + bytes([0x09, 0x02])) == b"abababab"
    # 2-byte-offset copy: literal "xyz" + copy(offset 3, len 3)
    assert tfc._snappy_decompress(bytes([6, 0x08]) + b"xyz" + bytes([(3 - 1) << 2 | 2, 3, 0])) == b"xyzxyz"
    with pytest.raises(ValueError):
        tfc._snappy_decompress(bytes([4, 0x09, 0x02]))                                   # copy before any output


def test_varints_and_negative_int64():
    for v in (0, 1, 127, 128, 300, 2 ** 32, 2 ** 63 - 1):
        enc = tfc._write_varint(v)
        assert tfc._read_varint(enc, 0) == (v, len(enc))
    enc = tfc._write_varint(-1)
    assert len(enc) == 10 and tfc._signed64(tfc._read_varint(enc, 0)[0]) == -1


def test_table_round_trip_many_blocks(tmp_path):
    rng = np.random.default_rng(0)
    items = [(("scope_%03d/kernel/part_%d" % (i // 3, i % 3)).encode(), rng.bytes(int(rng.integers(0, 200)))) for i in range(500)]
    path = str(tmp_path / "t.index")
    tfc.write_table(path, items, block_size=512)
    got = tfc.read_table(path)
    assert got == sorted(items)
    raw = bytearray(open(path, "rb").read())
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) > 48
    raw[10] ^= 0x40                                                                       # flip a bit in a data block
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_table(path)
    assert len(tfc.read_table(path, verify=False)) == 500                                # still parseable unverified


@pytest.mark.parametrize("net_work", ["FullyCNN", "FullyCNNV2", "FullyCNNV3"])
def test_checkpoint_round_trip_all_nets(tmp_path, net_work):
    variant = spec.variant_of(net_work)
    w = rced_np.make_weights(net_work, seed=11)
    extra = dict(w)
    extra["global_step"] = np.asarray(1234, np.int64)                                     # Saver stores these too
    extra["beta1_power"] = np.asarray(0.9 ** 7, np.float32)
    first = spec.variable_shapes(variant)[0][0]
    extra[first + "/Adam"] = np.zeros_like(w[first])
    prefix = str(tmp_path / "ckpt" / ("RCED_%s_0_9" % net_work))
    tfc.write_checkpoint(prefix, extra)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    got = W.load(prefix, variant)
    assert set(got) == set(w)
    for k in w:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], w[k])
    assert np.array_equal(W.pack_blob(variant, got), W.pack_blob(variant, w))
    # any file of the bundle names the checkpoint
    assert np.array_equal(W.load(prefix + ".index", variant)[first], w[first])
    allv = tfc.read_checkpoint(prefix + ".data-00000-of-00001")
    assert allv["global_step"].shape == () and int(allv["global_step"]) == 1234 and allv["global_step"].dtype == np.int64
    assert allv["beta1_power"].shape == ()


def test_checkpoint_errors(tmp_path):
    variant = spec.V3
    w = rced_np.make_weights("FullyCNNV3", seed=1)
    prefix = str(tmp_path / "m")
    tfc.write_checkpoint(prefix, w)
    with pytest.raises(KeyError, match="wrong net_work"):
        W.load(prefix, spec.V1)                                                           # V3 checkpoint into a V1 graph
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[100] ^= 1
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="data checksum"):
        W.load(prefix, variant)
    with pytest.raises(FileNotFoundError):
        W.load(str(tmp_path / "nothing"), variant)
    open(str(tmp_path / "junk.index"), "wb").write(b"not a table" * 10)
    with pytest.raises(ValueError, match="bad table magic"):
        tfc.read_checkpoint(str(tmp_path / "junk"))


def test_frozen_graph_round_trip_and_splat(tmp_path):
    w = rced_np.make_weights("FullyCNNV2", seed=5)
    pb = str(tmp_path / "RCED_FullyCNNV2_0_9.pb")
    tfc.write_frozen_graph(pb, w)
    got = W.load(pb, spec.V2)
    for k in w:
        assert np.array_equal(got[k], w[k])
    # a TensorProto holding one float_val for a whole tensor (how TF stores constants such as zeros)
    tp = tfc._field(1, 0, 1) + tfc._field(2, 2, tfc._shape_proto((2, 3))) + tfc._field(5, 2, struct.pack("<f", 0.25))
    assert np.array_equal(tfc._parse_tensor_proto(tp), np.full((2, 3), 0.25, np.float32))
    tp = tfc._field(1, 0, 3) + tfc._field(2, 2, tfc._shape_proto((3,))) + tfc._field(7, 2, b"".join(tfc._write_varint(v) for v in (1, 2, 300)))
    assert np.array_equal(tfc._parse_tensor_proto(tp), np.asarray([1, 2, 300], np.int32))


def test_engine_restores_from_checkpoint_prefix(tmp_path, monkeypatch):
    """tester.py:36-39: the engine's checkpoint_file may be a TF checkpoint prefix (weights only checked here;
    the forward itself needs the GPU and is covered by the gpu tests)."""
    from fullycnnspeechenhancement_amd import engine, model
    w = rced_np.make_weights("FullyCNNV3", seed=2)
    prefix = str(tmp_path / "RCED_FullyCNNV3_0_1")
    tfc.write_checkpoint(prefix, w)
    seen = {}
    monkeypatch.setattr(model.FullyCNNSEModelV3, "restore", lambda self, weights: seen.update(weights), raising=True)
    monkeypatch.setattr(model.FullyCNNSEModelV3, "__init__", lambda self, *a, **k: None, raising=True)
    eng = engine.FullyCNNTester(net_work="FullyCNNV3", checkpoint_file=prefix)
    assert eng.checkpoint_file == prefix and set(seen) == set(w)
    assert all(np.array_equal(seen[k], w[k]) for k in w)


def test_adam_step_is_recovered_past_the_underflow_of_beta1_power(tmp_path):
    """tf.train.AdamOptimizer keeps beta1^(t+1) in a float32: denormal near t = 830, zero near t = 980.  A checkpoint
    without global_step resumes at the count beta2_power (0.999^(t+1)) still gives; with neither, a warning says that
    the step is lost instead of silently restarting Adam's bias correction and the Noam schedule at 0."""
    import warnings
    w = rced_np.make_weights("FullyCNNV3", seed=2)
    for t, expect in ((10, 10), (500, 500), (900, 900), (2000, 2000)):
        extra = dict(w)
        extra["beta1_power"] = np.asarray(np.float32(0.9) ** np.float32(t + 1), np.float32)      # what TF holds (underflows)
        extra["beta2_power"] = np.asarray(0.999 ** (t + 1), np.float32)
        prefix = str(tmp_path / ("late_%d" % t))
        tfc.write_checkpoint(prefix, extra)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            _, _, _, step = tfc.load_training_state(prefix, spec.V3)
        assert step == expect, (t, step, float(extra["beta1_power"]))
    extra = dict(w)
    extra["beta1_power"] = np.asarray(0.0, np.float32)
    extra["beta2_power"] = np.asarray(0.0, np.float32)
    prefix = str(tmp_path / "lost")
    tfc.write_checkpoint(prefix, extra)
    with pytest.warns(UserWarning, match="cannot be recovered"):
        _, _, _, step = tfc.load_training_state(prefix, spec.V3)
    assert step == 0


def test_adam_step_from_powers_formed_the_way_tensorflow_forms_them(tmp_path):
    """TF builds beta1_power / beta2_power by REPEATED float32 multiplication with float32(0.9) / float32(0.999) (one per
    update).  At t ~ 5e4 the count is recovered from such a beta2_power to within one step (with log(0.999) as the base it
    was a step off from t = 40 k on), a checkpoint that also holds global_step loads without a spurious warning, and a real
    disagreement is reported under the name of the accumulator that was used."""
    import warnings
    w = rced_np.make_weights("FullyCNNV3", seed=2)
    t = 50000
    b1, b2 = np.float32(0.9), np.float32(0.999)
    p1, p2 = np.float32(1.0), np.float32(1.0)
    for _ in range(t + 1):
        p1 = np.float32(p1 * b1)
        p2 = np.float32(p2 * b2)
    assert p1 < 1e-30 and 1e-30 < p2 < 1e-20          # beta1_power is long gone (a denormal that 0.9 no longer moves)
    extra = dict(w, beta1_power=np.asarray(p1), beta2_power=np.asarray(p2))
    prefix = str(tmp_path / "no_step")
    tfc.write_checkpoint(prefix, extra)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, _, _, step = tfc.load_training_state(prefix, spec.V3)
    assert abs(step - t) <= 1, step
    extra["global_step"] = np.asarray(t, np.int64)
    prefix = str(tmp_path / "with_step")
    tfc.write_checkpoint(prefix, extra)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # agreement to within rounding: no warning
        _, _, _, step = tfc.load_training_state(prefix, spec.V3)
    assert step == t
    extra["global_step"] = np.asarray(t + 500, np.int64)
    prefix = str(tmp_path / "disagree")
    tfc.write_checkpoint(prefix, extra)
    with pytest.warns(UserWarning, match="beta2_power implies"):
        _, _, _, step = tfc.load_training_state(prefix, spec.V3)
    assert step == t + 500
