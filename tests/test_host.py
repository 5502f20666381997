"""CPU tests of the host-side mirror (weights container, shape checks, engine config parsing)."""

import configparser

import os

import numpy as np
import pytest

from conftest import NETS, load_golden
from oracle import rced_c, layers as L


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_blob_packing_matches_oracle_packing(net_work, tag, variant, built):
    from fullycnnspeechenhancement_amd import weights
    w, _ = load_golden(tag)
    mine = weights.pack_blob(variant, w)
    ref = rced_c.pack_blob(L.layers_for(net_work), w)
    assert mine.dtype == np.float32 and np.array_equal(mine, ref)


def test_initial_weights_follow_tf_defaults(built):
    from fullycnnspeechenhancement_amd import weights, spec
    w = weights.initial_weights(3, seed=0)
    assert set(w) == {n for n, _ in spec.variable_shapes(3)}
    k = w["CE1_encode_1/kernel"]
    lim = np.sqrt(6.0 / (8 * 9 * 1 + 8 * 9 * 18))
    assert k.shape == (8, 9, 1, 18) and np.abs(k).max() <= lim and np.abs(k).max() > 0.8 * lim
    assert not w["CE1_encode_1/bias"].any()
    assert (w["CE1_encode_1/batch_norm/gamma"] == 1).all() and (w["CE1_encode_1/batch_norm/moving_variance"] == 1).all()
    assert "decode_final/batch_norm/gamma" not in w     # use_norm=False on the last layer (model.py:89-90)


def test_validate_rejects_bad_weights(built):
    from fullycnnspeechenhancement_amd import weights
    w = weights.initial_weights(1, seed=0)
    bad = dict(w)
    del bad["encode_8/bias"]                        # V1's fifth encoder scope is "encode_8" (model.py:15)
    with pytest.raises(KeyError):
        weights.pack_blob(1, bad)
    bad = dict(w)
    bad["decode_5/kernel"] = np.zeros((1, 129, 12, 2), np.float32)
    with pytest.raises(ValueError):
        weights.pack_blob(1, bad)
    bad = dict(w)
    bad["encode_1/bias"] = np.full(12, np.nan, np.float32)
    with pytest.raises(ValueError):
        weights.pack_blob(1, bad)


def test_npz_round_trip(tmp_path, built):
    from fullycnnspeechenhancement_amd import weights
    w = weights.initial_weights(2, seed=3)
    p = str(tmp_path / "ckpt.npz")
    weights.save_npz(p, w)
    w2 = weights.load_npz(p)
    assert np.array_equal(weights.pack_blob(2, w), weights.pack_blob(2, w2))


def test_bench_gpus_n_without_a_launcher_starts_n_ranks(built):
    """`python bench.py --gpus 2` with RANK unset must start 2 ranks as a child (torch.distributed.run) and hand back
    the child's exit code.  Here there is no GPU, so each rank stops at bench.py's own "needs a GPU" check: what is
    checked is that two ranks with WORLD_SIZE=2 ran, and that the parent relayed their failure."""
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also on a GPU box: this test is about the launcher only
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=300)
    # What is deterministic: the parent relays the child's failure, no JSON line is invented, and at least one rank saw
    # WORLD_SIZE=2 (every rank prints that line BEFORE the GPU check; torchrun kills the surviving rank as soon as the
    # first one exits, so how many of the later "needs a GPU" messages appear is a race and is not asserted).
    assert r.returncode != 0
    assert "of WORLD_SIZE=2 started" in r.stderr, r.stderr[-2000:]
    assert "bench.py needs a GPU" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""            # no JSON line was produced, none was invented


def test_bench_rejects_a_world_size_that_is_not_gpus(built):
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ("needs a GPU" in r.stderr or "WORLD_SIZE" in r.stderr)


def test_shape_check_messages(built):
    from fullycnnspeechenhancement_amd.model import _RcedNet
    _RcedNet._check_shape((4, 7, 129, 1))
    for bad in ((4, 7, 129), (4, 7, 128, 1), (4, 129, 7, 1), (4, 7, 129, 2)):
        with pytest.raises(ValueError):
            _RcedNet._check_shape(bad)


def test_package_synthetic_weights_match_the_oracle_generator():
    """bench.py draws its random weights from the package (the product path never imports oracle/); the oracle's
    generator, which made the golden vectors, must stay the same recipe."""
    from fullycnnspeechenhancement_amd import spec, weights
    from oracle import rced_np
    for net_work in ("FullyCNN", "FullyCNNV2", "FullyCNNV3"):
        a = weights.synthetic_weights(spec.variant_of(net_work), seed=42)
        b = rced_np.make_weights(net_work, seed=42)
        assert set(a) == set(b)
        for k in a:
            assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k]), k


def test_compiled_kernels_are_free_of_the_two_measured_hazard_sequences(built):
    """tools/isa_lint.py over the gfx950 code of every object of the library: (A) VALU writes VCC / vector-memory
    instruction / SALU reads VCC, (B) an SGPR-base LDS-DMA closer than 5 wait states behind the VALU write of its base.
    Both compile without complaint and both misbehaved on MI355X; the sources avoid them with explicit wait
    states, and this test notices when a new build brings one back.  Also checks that the lint itself still sees them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import isa_lint
    bad_a = ["v_cmp_gt_i32_e32 vcc, 24, v22", "buffer_store_dwordx4 v[30:33], v24, s[44:47], s78 offen",
             "s_and_saveexec_b64 s[0:1], vcc"]
    assert [f[0] for f in isa_lint.lint_function("f", bad_a)] == ["A"]
    assert not isa_lint.lint_function("f", bad_a[:2] + ["s_nop 0"] + bad_a[2:])
    bad_b = ["v_readlane_b32 s21, v95, 3", "s_mov_b32 s16, m0", "s_mov_b32 m0, s3", "s_nop 0",
             "global_load_lds_dwordx4 v71, s[20:21]"]
    assert [f[0] for f in isa_lint.lint_function("f", bad_b)] == ["B"]
    assert not isa_lint.lint_function("f", bad_b[:3] + ["s_nop 3"] + bad_b[4:])
    bad_a2 = ["v_cmp_gt_i32_e64 s[0:1], 24, v22", "global_store_dwordx4 v[30:31], v[2:5], off", "s_and_saveexec_b64 s[2:3], s[0:1]"]
    assert [f[0] for f in isa_lint.lint_function("f", bad_a2)] == ["A"]
    assert not isa_lint.lint_function("f", bad_a2[:2] + ["s_and_saveexec_b64 s[2:3], s[4:5]"])
    objs = isa_lint.default_objects()     # the .so the package loads: what runs, not build/*.o that may be stale
    assert objs, "librced_hip.so exists after __graft_entry__.build()"
    found, nfn, nins = isa_lint.lint(objs)
    assert nfn > 100 and nins > 100000, (nfn, nins)
    assert not found, found[:3]


def test_host_output_pool_hands_out_fresh_looking_arrays():
    """model.HostOutputPool, the default of the numpy boundary (tester.py:85-90 returns a new ndarray per call): every take() is a new
    ndarray object; pages are recycled only when the caller holds nothing of the earlier result any more -- a slice counts; results that
    are kept are never written again; past the cap plain numpy.empty."""
    import numpy as np
    from fullycnnspeechenhancement_amd.model import HostOutputPool
    pool = HostOutputPool(max_bytes=3 * 4 * 24, per_shape=4)
    a = pool.take((2, 3, 4))
    a[...] = 1.0
    pa = a.__array_interface__["data"][0]
    b = pool.take((2, 3, 4))
    assert b is not a and b.__array_interface__["data"][0] != pa          # `a` is held: other pages
    keep = a[0, 1]                                                        # a slice keeps a's pages alive too
    del a
    c = pool.take((2, 3, 4))
    assert c.__array_interface__["data"][0] != pa and float(keep[0]) == 1.0
    del keep
    d = pool.take((2, 3, 4))
    assert d.__array_interface__["data"][0] == pa                         # dropped: the warm pages come back
    e = pool.take((2, 3, 4))                                              # b, c, d held, the cap (three buffers) reached: numpy.empty
    assert e.base is None and e.shape == (2, 3, 4) and e.dtype == np.float32
    assert pool.take((5, 1, 129, 1)).shape == (5, 1, 129, 1)
