"""Lightning-free counterpart of /root/reference/train.py for the hot path:  ``python -m a2f_amd.train --config config.yaml``.

Builds the model / loss from the same config.yaml, runs ``TrainStep`` (fwd + loss + bwd + all-reduce + Adam, weight decay
lr/10 as lightning_model.py:209-213) on synthetic VOCASET-shaped batches (the dataset is not part of this repository),
tracks the reference's ``err`` metric (mse_error, lightning_model.py:119-125) and saves a checkpoint whose keys carry the
reference's ``model.`` prefix.  Launch with torchrun for data parallel (one process per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.distributed as dist

from .registry import ExpConfig, build_from_config
from .trainer import TrainStep


def synthetic_batch(cfg: ExpConfig, batch: int, device, seed: int):
    g = torch.Generator().manual_seed(seed)
    nv = cfg.vertex_count // 3
    one_hot = torch.zeros(batch, cfg.one_hot_size)
    one_hot[torch.arange(batch), torch.randint(0, min(8, cfg.one_hot_size), (batch,), generator=g)] = 1.0
    template = torch.randn(batch, nv, 3, generator=g) * 0.05
    verts = template + torch.randn(batch, nv, 3, generator=g) * 0.005
    if cfg.modelname == "voca":
        audio = torch.randn(batch, 29, 16, generator=g)
    elif cfg.modelname == "audio2mesh":
        audio = torch.randn(batch, cfg.out_dim, cfg.n_feature, generator=g)
    else:
        raise NotImplementedError("use tools/bench_models.py for FaceFormer sequences")
    return {"audio": audio.to(device), "one_hot": one_hot.to(device), "verts": verts.to(device), "template_vert": template.to(device)}


def main(argv=None):
    # Lightning's Trainer(benchmark=None) turns cudnn.benchmark on when training is not deterministic (train.py:46-60): MIOpen then
    # autotunes every convolution shape once - the wav2vec2 feature encoder gains 8 % per FaceFormer step, backward included
    torch.backends.cudnn.benchmark = True
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="config.yaml")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args(argv)
    cfg = ExpConfig.from_yaml(args.config)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(0)
    model, loss_fn = build_from_config(cfg)
    model.to(device)
    step = TrainStep(model, loss_fn, lr=cfg.lr)
    batch_size = cfg.batch_size + (cfg.batch_size % 2)
    for it in range(args.steps):
        batch = synthetic_batch(cfg, batch_size, device, seed=1234 + rank + 1000 * it)
        x, one_hot, gt, template = TrainStep.unpack_batch(batch)          # x100 scaling, lightning_model.py:145-148
        out = step(x, one_hot, gt, template)
        if rank == 0 and it % 10 == 0:
            print(f"step {it}: loss {float(out['loss']):.5f} rec {float(out['rec_loss']):.5f} vel {float(out['vel_loss']):.5f} "
                  f"err {float(out['err']):.5f}", flush=True)
    if rank == 0 and args.out:
        torch.save({"state_dict": {"model." + k: v.detach().cpu() for k, v in model.state_dict().items()}, "config": cfg.model_dump()}, args.out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
