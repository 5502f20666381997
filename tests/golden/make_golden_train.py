"""Train-step golden vector (SURVEY 8(c) c4 iv): CR-CED V3 at [4,16,129,1], from THIS repo's fp64
PyTorch-autograd restatement (oracle/train_ref.py) -- NOT from TensorFlow, which cannot run here, so the
training parity is "unpinned" like the CNN forward.  Stored: inputs, loss of steps 1 and 2, gradients of the
first and last layers at step 1, and the variables after 1 and 2 Adam steps (lr 1e-3 then Noam).
Run from the repo root:  python tests/golden/make_golden_train.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rced_np, train_ref  # noqa: E402


def main():
    nw = "FullyCNNV3"
    w = rced_np.make_weights(nw, seed=42)
    x = rced_np.make_input(4, 16, seed=1234)
    y = rced_np.make_input(4, 16, seed=1235)
    tr = train_ref.TrainRef(nw, w, batch_size=4)
    out = {"x": x, "y": y}
    lr0, warm = 1e-3, 4000.0
    loss1, grads, _ = tr.loss_and_grads(x, y)
    for k in ("CE1_encode_1/kernel", "CE1_encode_1/batch_norm/gamma", "decode_final/kernel", "decode_final/bias"):
        out["g1:" + k] = grads[k].numpy()
    l1, s1 = tr.train_step(x, y, lr0)
    assert abs(l1 - loss1) < 1e-9
    for k, v in tr.weights().items():
        out["v1:" + k] = v
    lr1 = train_ref.noam_lr(lr0, s1, warm)
    l2, s2 = tr.train_step(x, y, lr1)
    for k, v in tr.weights().items():
        out["v2:" + k] = v
    out["loss"] = np.asarray([l1, l2])
    out["lr"] = np.asarray([lr0, lr1])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_v3.npz"), **out)
    print("loss", l1, l2, "lr", lr0, lr1)


if __name__ == "__main__":
    main()
