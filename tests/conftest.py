import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """The HIP library and the C oracle, built in-tree (no-op when up to date)."""
    import __graft_entry__ as ge
    ge.build_hip()
    ge.build_oracle()
    return True


def load_golden(tag):
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "rced_%s.npz" % tag))
    w = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    return w, {k: z[k] for k in z.files if not k.startswith("w:")}


NETS = [("FullyCNN", "v1", 1), ("FullyCNNV2", "v2", 2), ("FullyCNNV3", "v3", 3)]
# floating-point bar of BASELINE.json north_star: masks within 1e-4 relative (fp32)
RTOL = 1e-4


def rel_err(y, ref):
    import numpy as np
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(y, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


def check_parity(y, ref, rtol=RTOL, what=""):
    """The fp32 bar, two ways: max|y - ref| <= rtol * max|ref| (the north star's "1e-4 rel" read against the mask's
    scale) AND element-wise |y - ref| <= rtol * |ref| + 0.1 * rtol * max|ref| (numpy.allclose with an absolute floor of
    1e-5 of the scale, so that entries near zero are not asked for more digits than fp32 summation has).
    Returns the measured max error relative to the scale."""
    import numpy as np
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(y - ref).max() / scale)
    assert err < rtol, "%s max error %.3e of the scale (bar %.1e)" % (what, err, rtol)
    bad = np.abs(y - ref) > rtol * np.abs(ref) + 0.1 * rtol * scale
    assert not bad.any(), "%s %d entries outside rtol %.1e / atol %.1e*scale" % (what, int(bad.sum()), rtol, 0.1 * rtol)
    return err
