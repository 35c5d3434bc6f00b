"""Config surface and registries of the reference training harness (reference:
/root/reference/src/model/lightning_model.py:20-73 and /root/reference/config.yaml), Lightning-free.

``ExpConfig`` keeps every key of config.yaml - including the reference's spelling ``percision`` - and
``get_model`` / ``get_loss_fn`` return the HIP-backed drop-in classes for the three models on the hot path.
Song2Face / af_model are not on the accelerated path and raise a clear error.
"""
from __future__ import annotations

from typing import Optional

import yaml
from pydantic import BaseModel

from .loss import FaceFormerLoss, VocaLoss
from .model.audio2face import Audio2Mesh
from .model.faceformer import Faceformer
from .model.voca import Voca


class ExpConfig(BaseModel):
    # dataset
    batch_size: int
    # model
    modelname: str
    one_hot_size: int
    feature_extractor: Optional[str] = None
    sample_rate: int
    vertex_count: int
    split_frame: bool
    n_feature: int
    out_dim: int
    win_length: int
    hop_length: Optional[int] = None
    # training
    percision: str
    lr: float
    # loss
    loss: Optional[str] = None

    @classmethod
    def from_yaml(cls, path: str):
        with open(path, "r") as f:
            return cls(**yaml.safe_load(f))

    def name(self):
        return f"{self.modelname}_{self.feature_extractor}_{self.lr}_{self.loss}_{self.percision}"


_MODELS = {"voca": Voca, "audio2mesh": Audio2Mesh, "faceformer": Faceformer}


def get_model(modelname: str):
    try:
        return _MODELS[modelname]
    except KeyError:
        raise KeyError(f"model '{modelname}' is not on the MI355X hot path (available: {sorted(_MODELS)}); "
                       f"use the reference implementation for it") from None


def get_loss_fn(modelname: str):
    return FaceFormerLoss() if modelname == "faceformer" else VocaLoss()


def get_extractor(name: Optional[str]):
    """lightning_model.py:61-67; None -> no extractor (features are fed directly, lightning_model.py:112-113)."""
    if name is None:
        return None
    from .extractor import get_extractor as _get
    return _get(name)


def build_extractor(config: ExpConfig):
    """The feature extractor exactly as Audio2FaceModel.__init__ constructs it (lightning_model.py:84-93), or None."""
    fe = get_extractor(config.feature_extractor)
    if fe is None:
        return None
    return fe(sample_rate=config.sample_rate, n_feature=config.n_feature, out_dim=config.out_dim, win_length=config.win_length,
              hop_length=config.hop_length, n_fft=1024)


def build_from_config(config: ExpConfig):
    """(model, loss_fn) exactly as Audio2FaceModel.__init__ builds them (lightning_model.py:96-99)."""
    model = get_model(config.modelname)(n_verts=config.vertex_count, n_onehot=config.one_hot_size)
    # `percision` (the reference's spelling) is Lightning's precision flag: "16-mixed" / "bf16-mixed" autocast the torch-op
    # encoder of FaceFormer; everything that runs in the HIP library is fp32 storage with fp32 accumulation regardless.
    amp = mixed_precision_dtype(config.percision)
    if amp is not None and hasattr(model, "encoder_autocast"):
        model.encoder_autocast = amp
    return model, get_loss_fn(config.modelname)


def mixed_precision_dtype(percision: str):
    """Lightning precision string -> autocast dtype (None for the 32-bit settings)."""
    import torch
    p = str(percision).lower()
    if p.startswith("bf16"):
        return torch.bfloat16
    if p.startswith("16"):
        # Lightning's "16-mixed" = fp16 autocast + GradScaler (dynamic loss scale, unscale, skip-on-inf).  TrainStep / FlatAdam
        # carry no loss scaling, so fp16 gradients could underflow or push an inf into Adam: the 16-bit setting runs as bf16
        # (fp32 exponent range, no scaler needed) and says so.
        import warnings
        warnings.warn('percision "16-mixed": no GradScaler in this trainer - the encoder runs in bf16 mixed precision instead of fp16')
        return torch.bfloat16
    return None
