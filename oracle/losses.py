"""Losses and the step-side metric, CPU restatement (test infrastructure).

* ``voca_loss``        <- /root/reference/src/loss/loss.py:24-55
* ``faceformer_loss``  <- /root/reference/src/loss/loss.py:4-17
* ``mse_error``        <- /root/reference/src/model/lightning_model.py:119-125
"""
from __future__ import annotations

from typing import Dict

import torch


def voca_loss(pred: torch.Tensor, gt: torch.Tensor, k_rec: float = 1.0, k_vel: float = 10.0) -> Dict[str, torch.Tensor]:
    bs = pred.shape[0]
    p = pred.reshape(bs, -1, 3)
    g = gt.reshape(bs, -1, 3)
    nv = p.shape[1]
    rec = ((p - g) ** 2).sum(dim=2).mean()                       # loss.py:29-30
    pp = p.reshape(-1, 2, nv, 3)                                 # loss.py:33-35  pairs rows (2k,2k+1); needs even bs
    gg = g.reshape(-1, 2, nv, 3)
    dv = (pp[:, 1] - pp[:, 0]) - (gg[:, 1] - gg[:, 0])           # loss.py:37-40
    vel = (dv ** 2).sum(dim=2).mean()
    return {"loss": rec * k_rec + vel * k_vel, "rec_loss": rec, "vel_loss": vel}   # loss.py:51-55


def faceformer_loss(pred: torch.Tensor, gt: torch.Tensor) -> Dict[str, torch.Tensor]:
    g = gt.squeeze(0)                                            # loss.py:10-11
    p = pred.squeeze(0)
    if g.shape[0] % 2 != 0:                                      # loss.py:13-15 drop odd tail frame
        g, p = g[:-1], p[:-1]
    return voca_loss(p, g)


def mse_error(pred: torch.Tensor, gt: torch.Tensor, width: int = 5023 * 3) -> torch.Tensor:
    with torch.no_grad():
        d = pred.reshape(-1, width) - gt.reshape(-1, width)
        return (d ** 2).mean(dim=1).mean()
