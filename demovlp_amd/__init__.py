"""demovlp_amd: MI355X (gfx950) native implementation of DemoVLP's cross-modal forward/backward hot path.

Module names mirror the reference (model/model.py -> demovlp_amd.model, model/loss.py -> demovlp_amd.loss,
model/object_transformer.py -> demovlp_amd.object_transformer, trainer/trainer_dist.py -> demovlp_amd.trainer) so a
DemoVLP ``ConfigParser.initialize('arch', demovlp_amd.model)`` call builds the drop-in module from the same JSON.
"""
from ._lib import DemoVLPHipError, LIB_PATH  # noqa: F401

__all__ = ["DemoVLPHipError", "LIB_PATH"]
