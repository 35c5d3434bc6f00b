"""a2f_amd: MI355X-native hot path of xtliu97/audio2face-pytorch.

Import as ``import a2f_amd`` (the repo-root shim loads this hyphen-named directory).
"""
from ._lib import A2FError                      # noqa: F401
from .loss import FaceFormerLoss, VocaLoss      # noqa: F401
from .model.audio2face import Audio2Mesh        # noqa: F401
from .model.faceformer import Faceformer        # noqa: F401
from .model.voca import Voca                    # noqa: F401

__all__ = ["Voca", "Audio2Mesh", "Faceformer", "VocaLoss", "FaceFormerLoss", "A2FError"]
