"""Lightning-free counterpart of the reference's fit loop (reference: /root/reference/train.py:28-67 and
/root/reference/src/model/lightning_model.py:127-207): epochs of ``TrainStep`` over a training loader, a validation pass per
epoch that produces ``val/err`` (epoch mean of ``mse_error``, lightning_model.py:119-125,135-143,163-173), the two callbacks
train.py configures - ``ModelCheckpoint(monitor="val/err", save_last=False)`` (best checkpoint only, named
``epoch=E-step=S.ckpt``) and ``EarlyStopping(monitor="val/err", patience=5)`` - ``max_epochs=50``, and the reload of the
checkpoint with the largest ``step=`` in its file name (train.py:62-67).

Checkpoints are dictionaries in Lightning's layout: ``state_dict`` with the reference's ``model.`` prefix (``Audio2FaceModel``
holds the network as ``self.model``), ``hyper_parameters = {"config": ExpConfig}`` (what ``save_hyperparameters()`` records,
lightning_model.py:109), ``epoch`` / ``global_step`` / ``pytorch-lightning_version``, plus ``optimizer_states`` (the flat Adam
moments and step count) so training can resume - the reference has no resume path of its own.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional

import torch

from .registry import ExpConfig, build_from_config
from .trainer import TrainStep

LIGHTNING_VERSION_TAG = "2.0.0"          # written as `pytorch-lightning_version`; Lightning only reads it to pick migrations


def apply_model_overrides(config: ExpConfig) -> ExpConfig:
    """train.py:30-34: FaceFormer trains whole sequences, batch 1, raw audio (no feature extractor)."""
    if config.modelname == "faceformer":
        config.split_frame = False
        config.batch_size = 1
        config.feature_extractor = None
    return config


class EarlyStopping:
    """lightning.pytorch.callbacks.EarlyStopping(monitor, patience) with its defaults (mode="min", min_delta=0, strict): the run
    stops after `patience` consecutive validation epochs without an improvement of the monitored value."""

    def __init__(self, patience: int = 5, min_delta: float = 0.0):
        self.patience, self.min_delta = patience, min_delta
        self.best = float("inf")
        self.wait = 0
        self.stopped_epoch: Optional[int] = None

    def update(self, value: float, epoch: int) -> bool:
        if value != value:                       # NaN: Lightning's check_finite stops the run
            self.stopped_epoch = epoch
            return True
        if value < self.best - self.min_delta:
            self.best, self.wait = value, 0
            return False
        self.wait += 1
        if self.wait >= self.patience:
            self.stopped_epoch = epoch
            return True
        return False


class ModelCheckpoint:
    """ModelCheckpoint(monitor="val/err", save_last=False): save_top_k = 1, mode = "min" - keep only the best checkpoint."""

    def __init__(self, dirpath: str):
        self.dirpath = dirpath
        self.best = float("inf")
        self.best_path: Optional[str] = None

    def update(self, value: float, epoch: int, global_step: int, save_fn: Callable[[str], None]) -> bool:
        if not value < self.best:
            return False
        os.makedirs(self.dirpath, exist_ok=True)
        path = os.path.join(self.dirpath, f"epoch={epoch}-step={global_step}.ckpt")
        save_fn(path)
        if self.best_path and self.best_path != path and os.path.exists(self.best_path):
            os.remove(self.best_path)
        self.best, self.best_path = value, path
        return True


def save_checkpoint(path: str, model: torch.nn.Module, step: Optional[TrainStep], config: ExpConfig, epoch: int, global_step: int):
    ckpt = {
        "epoch": epoch, "global_step": global_step, "pytorch-lightning_version": LIGHTNING_VERSION_TAG,
        "state_dict": {"model." + k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "hyper_parameters": {"config": config},
    }
    if step is not None:
        o = step.opt
        ckpt["optimizer_states"] = [{"kind": "a2f_amd.FlatAdam", "step_count": o.step_count, "lr": o.lr, "weight_decay": o.wd, "betas": o.betas,
                                     "eps": o.eps, "m": o.m.detach().cpu().clone(), "v": o.v.detach().cpu().clone()}]
    torch.save(ckpt, path)


def load_checkpoint(path: str, model: torch.nn.Module, step: Optional[TrainStep] = None) -> Dict:
    """Restores the network (keys with the ``model.`` prefix; ``feature_extractor.*`` keys of a Lightning checkpoint are skipped)
    and, when given, the optimizer state of ``step``.  Returns the checkpoint dictionary."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
    model.load_state_dict(sd)
    if step is not None and ckpt.get("optimizer_states"):
        st = ckpt["optimizer_states"][0]
        if st.get("kind") == "a2f_amd.FlatAdam":
            step.opt.m.copy_(st["m"]); step.opt.v.copy_(st["v"])
            step.opt.step_count = int(st["step_count"])
    return ckpt


def best_checkpoint(dirpath: str) -> str:
    """train.py:62-63: the checkpoint whose file name carries the largest ``step=``."""
    ckpts = [f for f in os.listdir(dirpath) if f.endswith(".ckpt")]
    return os.path.join(dirpath, sorted(ckpts, key=lambda x: int(x.split("=")[-1].split(".")[0]))[-1])


@torch.no_grad()
def mse_error(pred: torch.Tensor, gt: torch.Tensor, width: int = 5023 * 3) -> torch.Tensor:
    """lightning_model.py:119-125 (the frame width is hard-coded there)."""
    return torch.mean(torch.mean((pred.reshape(-1, width) - gt.reshape(-1, width)) ** 2, dim=1))


def validate(model: torch.nn.Module, loss_fn, batches: Iterable[Dict[str, torch.Tensor]], extractor=None) -> Dict[str, float]:
    """validation_step + on_validation_epoch_end (lightning_model.py:135-143,163-173): eval mode, no gradients."""
    was_training = model.training
    model.eval()
    errs: List[float] = []
    last = {}
    with torch.no_grad():
        for batch in batches:
            x, one_hot, gt, template = TrainStep.unpack_batch(batch)
            if extractor is not None:
                x = extractor(x).detach()
            pred = model(x, one_hot, template)
            last = {k: float(v) for k, v in loss_fn(pred, gt).items()}
            errs.append(float(mse_error(pred, gt, width=pred.shape[-2] * pred.shape[-1])))
    model.train(was_training)
    return {"val/err": sum(errs) / max(len(errs), 1), **{"val/" + k: v for k, v in last.items()}}


def fit(config: ExpConfig, train_batches: Callable[[int], Iterable[Dict[str, torch.Tensor]]],
        val_batches: Callable[[int], Iterable[Dict[str, torch.Tensor]]], device, log_dir: str, max_epochs: int = 50, patience: int = 5,
        log: Callable[[str], None] = print, model: Optional[torch.nn.Module] = None, use_graph: bool = False):
    """trainer.fit + the reload of train.py:62-67.  ``train_batches(epoch)`` / ``val_batches(epoch)`` yield the batch dictionaries
    of the reference's data module (``audio``, ``one_hot``, ``verts``, ``template_vert``; un-scaled: the x100 happens here).
    Returns (model with the best checkpoint loaded, history)."""
    config = apply_model_overrides(config)
    loss_fn = None
    if model is None:
        model, loss_fn = build_from_config(config)
    if loss_fn is None:
        from .registry import get_loss_fn
        loss_fn = get_loss_fn(config.modelname)
    from .registry import build_extractor
    extractor = build_extractor(config)
    model.to(device)
    if extractor is not None:
        extractor.to(device)
    step = TrainStep(model, loss_fn, lr=config.lr, use_graph=use_graph)
    if step.opt.world() > 1:
        # identical initial weights on every rank (same seed before the model was built); from here on every rank draws its own
        # stochastic-regulariser masks (FaceFormer train mode), as DDP ranks do in the reference
        import torch.distributed as dist
        torch.manual_seed(torch.initial_seed() + 7919 * (dist.get_rank() + 1))
    ckpt_cb = ModelCheckpoint(os.path.join(log_dir, "checkpoints"))
    stop_cb = EarlyStopping(patience=patience)
    history = []
    global_step = 0
    for epoch in range(max_epochs):
        model.train()
        errs = []
        for batch in train_batches(epoch):
            x, one_hot, gt, template = TrainStep.unpack_batch(batch)
            if extractor is not None:
                x = extractor(x).detach()                       # lightning_model.py:115
            out = step(x, one_hot, gt, template)
            errs.append(float(out["err"]))                       # the reference's per-step .item() (lightning_model.py:160)
            global_step += 1
        rec = {"epoch": epoch, "global_step": global_step, "train/err": sum(errs) / max(len(errs), 1)}
        rec.update(validate(model, loss_fn, val_batches(epoch), extractor))
        rec["saved"] = ckpt_cb.update(rec["val/err"], epoch, global_step,
                                      lambda p: save_checkpoint(p, model, step, config, epoch, global_step))
        history.append(rec)
        log(f"Epoch {epoch} train err: {rec['train/err']:.6f}  val error: {rec['val/err']:.6f}" + ("  (checkpoint)" if rec["saved"] else ""))
        if stop_cb.update(rec["val/err"], epoch):
            log(f"early stop at epoch {epoch}: val/err has not improved for {patience} epochs (best {stop_cb.best:.6f})")
            break
    if ckpt_cb.best_path:
        load_checkpoint(best_checkpoint(ckpt_cb.dirpath), model)
    return model, history
