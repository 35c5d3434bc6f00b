"""Lightning-free counterpart of /root/reference/train.py for the hot path:  ``python -m a2f_amd.train --config config.yaml``.

Builds model / loss / feature extractor from the same config.yaml (FaceFormer overrides of train.py:30-34 included), runs the
fit loop of ``a2f_amd.fit`` - ``TrainStep`` (fwd + loss + bwd + all-reduce + Adam, weight decay lr/10 as
lightning_model.py:209-213), a validation pass per epoch, ``val/err`` best-checkpoint and early stopping (patience 5,
max_epochs 50) as train.py:48-60 configures them - on synthetic VOCASET-shaped batches (the dataset is not part of this
repository), then reloads the best checkpoint (train.py:62-67) and runs one prediction (x / 100 as predict_step does,
lightning_model.py:215-225).  Launch with torchrun for data parallel (one process per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.distributed as dist

from .fit import apply_model_overrides, fit
from .registry import ExpConfig


def synthetic_batch(cfg: ExpConfig, batch: int, device, seed: int, frames: int = 120):
    """One batch dictionary shaped like VocaDataModule's (vocaset.py:248-289): per-frame windows for VOCA / Audio2Mesh, one whole
    clip (batch 1) for FaceFormer.  Vertices in metres (the x100 happens in unpack_batch)."""
    g = torch.Generator().manual_seed(seed)
    nv = cfg.vertex_count // 3
    one_hot = torch.zeros(batch, cfg.one_hot_size)
    one_hot[torch.arange(batch), torch.randint(0, min(8, cfg.one_hot_size), (batch,), generator=g)] = 1.0
    template = torch.randn(batch, nv, 3, generator=g) * 0.05
    if cfg.modelname == "voca":
        audio = torch.randn(batch, 29, 16, generator=g)
        verts = template + torch.randn(batch, nv, 3, generator=g) * 0.005
    elif cfg.modelname == "audio2mesh":
        if cfg.feature_extractor is None:
            audio = torch.randn(batch, cfg.out_dim, cfg.n_feature, generator=g)
        else:
            audio = torch.randn(batch, int(0.52 * cfg.sample_rate), generator=g) * 0.1     # 0.52 s windows (vocaset.py:253-260)
        verts = template + torch.randn(batch, nv, 3, generator=g) * 0.005
    else:                                                                                # faceformer: (1, N) @ 16 kHz, (1, T, V, 3)
        audio = torch.randn(1, frames * 16000 // 60 + 8, generator=g) * 0.1
        one_hot, template = one_hot[:1], template[:1]
        verts = template.view(1, 1, nv, 3) + torch.randn(1, frames, nv, 3, generator=g) * 0.005
    return {"audio": audio.to(device), "one_hot": one_hot.to(device), "verts": verts.to(device), "template_vert": template.to(device)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="config.yaml")
    ap.add_argument("--max-epochs", type=int, default=50)
    ap.add_argument("--steps-per-epoch", type=int, default=20)
    ap.add_argument("--val-steps", type=int, default=4)
    ap.add_argument("--frames", type=int, default=120, help="FaceFormer: frames per synthetic clip")
    ap.add_argument("--log-dir", type=str, default="logs/a2f_amd")
    args = ap.parse_args(argv)
    cfg = apply_model_overrides(ExpConfig.from_yaml(args.config))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(0)
    batch_size = cfg.batch_size + (cfg.batch_size % 2 if cfg.modelname != "faceformer" else 0)      # VocaLoss pairs rows (2k, 2k+1)

    def train_batches(epoch):
        for it in range(args.steps_per_epoch):
            yield synthetic_batch(cfg, batch_size, device, seed=1234 + rank + 1000 * (epoch * args.steps_per_epoch + it), frames=args.frames)

    def val_batches(epoch):
        for it in range(args.val_steps):
            yield synthetic_batch(cfg, batch_size, device, seed=99_000 + it, frames=args.frames)

    log = print if rank == 0 else (lambda *_: None)
    model, history = fit(cfg, train_batches, val_batches, device, os.path.join(args.log_dir, cfg.name()), max_epochs=args.max_epochs, log=log)
    batch = synthetic_batch(cfg, batch_size, device, seed=7, frames=args.frames)
    from .trainer import TrainStep
    x, one_hot, gt, template = TrainStep.unpack_batch(batch)
    model.eval()
    ex = fit.last_extractor                                     # the extractor fit trained with (not a freshly initialised one), in eval mode
    if isinstance(ex, torch.nn.Module):
        ex.eval()
    with torch.no_grad():
        pred = model.predict(ex(x) if ex is not None else x, one_hot, template) / 100            # predict_step: back to metres
    log(f"predicted {tuple(pred.shape)} vertices (metres), best val/err {min(h['val/err'] for h in history):.6f} after {len(history)} epochs")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
