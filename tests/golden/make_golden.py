"""Generates the committed golden vectors under tests/golden/.

PROVENANCE: these come from THIS repo's fp64 restatement (oracle/rced_np.py), not from the
TensorFlow reference: TF 1.14 cannot be imported in the build container and the reference ships
no vectors -- parity stays "unpinned".  The vectors freeze the restatement so that
a later change to oracle, host code or kernels that moves results is caught.

Run from the repo root:  python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rced_np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for net_work, tag in (("FullyCNN", "v1"), ("FullyCNNV2", "v2"), ("FullyCNNV3", "v3")):
        w = rced_np.make_weights(net_work, seed=42)
        x_small = rced_np.make_input(2, 16, seed=1234)
        x_long = rced_np.make_input(1, 40, seed=4321)
        x_c1 = rced_np.make_input(1, 256, seed=1234)      # BASELINE configs[0]'s shape, [1,256,129,1] (SURVEY 8 c4 iii)
        out = {"x_small": x_small, "y_small": rced_np.forward(net_work, w, x_small),
               "x_long": x_long, "y_long": rced_np.forward(net_work, w, x_long),
               "x_c1": x_c1, "y_c1": rced_np.forward(net_work, w, x_c1)}
        for k, v in w.items():
            out["w:" + k] = v
        np.savez_compressed(os.path.join(HERE, "rced_%s.npz" % tag), **out)
        print(tag, {k: v.shape for k, v in out.items() if not k.startswith("w:")})


if __name__ == "__main__":
    main()
