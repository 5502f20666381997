"""fullycnnspeechenhancement_amd -- MI355X-native R-CED / CR-CED forward pass.

One hot path of phecda-xu/FullyCNNSpeechEnhancement (model_utils/model.py + module.py) as
hand-written gfx950 HIP kernels behind a C ABI (include/rced.h, librced_hip.so), with this
package as the Python host side that keeps the reference's `Model(is_training)(x)` /
`engine.test_step(ndarray)` surface.
"""

from .model import (FullyCNNSEModel, FullyCNNSEModelV2, FullyCNNSEModelV3, build_model,  # noqa: F401
                    conv_bn_relu)
from .engine import FullyCNNTester, InferenceEngine  # noqa: F401
from .trainer import FullyCNNTrainer  # noqa: F401
from . import audio, spec, weights  # noqa: F401

__all__ = ["FullyCNNSEModel", "FullyCNNSEModelV2", "FullyCNNSEModelV3", "build_model", "conv_bn_relu",
           "FullyCNNTester", "InferenceEngine", "FullyCNNTrainer", "audio", "spec", "weights"]
