"""Drop-in ``VocaLoss`` / ``FaceFormerLoss`` (reference: /root/reference/src/loss/loss.py:4-55).

Same call signature and the same dict of three scalars; the arithmetic is one streaming HIP kernel
forward and one backward (pred and gt are each read once per pass).
"""
from __future__ import annotations

from ..functional import VocaLossFn


class VocaLoss:
    def __init__(self, k_rec: float = 1.0, k_vel: float = 10.0):
        self.k_rec = k_rec
        self.k_vel = k_vel

    def __call__(self, pred, gt):
        bs = pred.shape[0]
        if bs % 2 != 0:
            raise RuntimeError(f"VocaLoss pairs consecutive frames (loss.py:33-35): batch {bs} must be even")
        p2 = pred.reshape(bs, -1)
        g2 = gt.reshape(bs, -1)
        self.n_verts = p2.shape[1] // 3
        out3 = VocaLossFn.apply(p2, g2, self.k_rec, self.k_vel)
        return {"loss": out3[0], "rec_loss": out3[1], "vel_loss": out3[2]}


class FaceFormerLoss:
    def __init__(self) -> None:
        self.loss = VocaLoss()

    def __call__(self, pred, gt):
        gt = gt.squeeze(0)
        pred = pred.squeeze(0)
        if gt.shape[0] % 2 != 0:          # loss.py:13-15: drop the last frame of an odd-length sequence
            gt = gt[:-1]
            pred = pred[:-1]
        return self.loss(pred, gt)
