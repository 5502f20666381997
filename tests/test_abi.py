"""CPU tests: the C-ABI library loads, exports every symbol include/rced.h declares, reports the
reference's topology, and fails loudly (no CPU fallback) when no GPU is present."""

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import NETS, ROOT
from oracle import layers as L


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rced.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rced_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    from fullycnnspeechenhancement_amd import _lib
    lib = ctypes.CDLL(_lib.SO_PATH)
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names          # the binding covers the header, nothing else
    assert "gfx950" in _lib.version()


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "rced.h")).read()
    assert "torch" not in src.lower() and "at::" not in src and "std::" not in src


@pytest.mark.parametrize("net_work,tag,variant", NETS)
def test_topology_matches_oracle_tables(net_work, tag, variant, built):
    from fullycnnspeechenhancement_amd import spec
    mine = spec.layers(variant)
    ref = L.layers_for(net_work)
    assert len(mine) == len(ref)
    for i, (a, b) in enumerate(zip(mine, ref)):
        assert (a.scope, a.cout, a.kh, a.kw, bool(a.use_norm), bool(a.use_act), a.src, a.skip_pre, a.skip_post) == \
               (b.scope, b.cout, b.kh, b.kw, b.use_norm, b.use_act, b.src, b.skip_pre, b.skip_post)
        assert a.cin == L.cin_of(ref, i)
    assert spec.num_trainable(variant) == L.param_count(ref)
    assert [n for n, _ in spec.variable_shapes(variant)] == [n for n, _ in L.variable_shapes(ref)]
    assert spec.variant_of(net_work) == variant


def test_published_counts_and_flops(built):
    from fullycnnspeechenhancement_amd import spec
    assert [spec.num_trainable(v) for v in (1, 2, 3)] == [32765, 32192, 32653]   # readme.md:65-67
    assert [spec.flops_per_frame(v) for v in (1, 2, 3)] == [8316888, 8109456, 8207496]
    assert spec.variant_of("FullyCNN") == 1 and spec.variant_of("typo") == 1      # infer.py:45-51 default


def test_bad_arguments_are_reported_not_crashed(built):
    from fullycnnspeechenhancement_amd import _lib
    lib = _lib.load()
    assert lib.rced_num_layers(7) == -1 and lib.rced_num_weights(0) == 0
    h = ctypes.c_void_p()
    blob = np.zeros(10, np.float32)
    rc = lib.rced_create(3, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), blob.size, 0, ctypes.byref(h))
    assert rc == _lib.RCED_ERR_ARG and b"needs 33213" in lib.rced_last_error()
    rc = lib.rced_create(9, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), blob.size, 0, ctypes.byref(h))
    assert rc == _lib.RCED_ERR_ARG
    assert lib.rced_forward(None, None, None, 1, 1, None) == _lib.RCED_ERR_ARG
    lib.rced_destroy(None)   # like free(NULL)


def test_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fullycnnspeechenhancement_amd import FullyCNNSEModelV3, _lib
    with pytest.raises(_lib.RcedError) as ei:
        FullyCNNSEModelV3(False)
    assert ei.value.code == _lib.RCED_ERR_HIP and "no CPU fallback" in str(ei.value)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fullycnnspeechenhancement_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                s = open(os.path.join(dp, f)).read()
                assert "oracle" not in s.replace("no oracle", ""), os.path.join(dp, f)
