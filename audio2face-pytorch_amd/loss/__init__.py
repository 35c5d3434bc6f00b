from .loss import FaceFormerLoss, VocaLoss

__all__ = ["VocaLoss", "FaceFormerLoss"]
