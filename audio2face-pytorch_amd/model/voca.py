"""Drop-in ``Voca`` (reference: /root/reference/src/model/voca.py:5-52) on the HIP hot path.

Same constructor signature, same sub-module names / parameter shapes / construction order (so seeded
initialisation and ``state_dict`` interchange with the reference), same ``forward(x, one_hot,
template, **kwargs)`` / ``predict`` surface.  The arithmetic is not torch.nn: forward and backward are
launches of liba2f_hip.so (implicit-GEMM temporal convolutions, fused Linear epilogues, fused head).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..functional import HeadFn, HeadLossFusedFn, VocaTrunkFn, VOCA_CONVS, voca_frames, voca_trunk_bwd, voca_trunk_fwd

_FC = ((72, 72), (72, 128), None, (128, 50))        # decoder slots 0,1,(2 = Tanh),3; slot 4 is the vertex head


class Voca(nn.Module):
    def __init__(self, n_verts: int, n_onehot: int):
        super().__init__()
        self.n_verts = n_verts
        self.n_onehot = n_onehot
        tc = []
        for ci, co, _ in VOCA_CONVS:                          # voca.py:19-28
            tc += [nn.Conv2d(ci, co, kernel_size=(3, 1), stride=(2, 1), padding=(1, 0)), nn.ReLU()]
        self.time_conv = nn.Sequential(*tc)
        dec = [nn.Tanh() if s is None else nn.Linear(*s) for s in _FC]
        dec.append(nn.Linear(50, n_verts))                    # voca.py:30-36
        self.decoder = nn.Sequential(*dec)

    def _trunk_params(self):
        ps = []
        for i in (0, 2, 4, 6):
            ps += [self.time_conv[i].weight, self.time_conv[i].bias]
        for i in (0, 1, 3):
            ps += [self.decoder[i].weight, self.decoder[i].bias]
        return ps

    def trunk(self, x, one_hot):
        """(B,29,16),(B,n_onehot) -> 50-wide bottleneck (B,50).  Extension: ``x`` may be the per-frame
        feature tensor (S,T,29) of S sequences; the 16-frame windows (vocaset.py zero-padded slices) are then built inside
        the trunk kernel and B = S*T."""
        return VocaTrunkFn.apply(x, one_hot, *self._trunk_params())

    def forward(self, x, one_hot, template, **kwargs):
        bs = voca_frames(x)
        h3 = self.trunk(x, one_hot)
        pred = HeadFn.apply(h3, self.decoder[4].weight, self.decoder[4].bias, template.reshape(bs, -1))
        return pred.view(bs, -1, 3)

    def predict(self, x, one_hot, template, **kwargs):
        return self(x, one_hot, template, **kwargs)

    def forward_with_loss(self, x, one_hot, template, gt, k_rec: float = 1.0, k_vel: float = 10.0, unit_grad: bool = False):
        """Training fast path: model forward + VocaLoss + head backward fused into one pass over the frame matrix.

        Returns (pred (B,V,3) detached, {"loss","rec_loss","vel_loss"}); only "loss" carries gradient.  Numerically the
        same function as ``VocaLoss(k_rec,k_vel)(self(x, one_hot, template), gt)``.
        """
        bs = voca_frames(x)
        if bs % 2 != 0:
            raise RuntimeError(f"VocaLoss pairs consecutive frames (loss.py:33-35): batch {bs} must be even")
        h3 = self.trunk(x, one_hot)
        pred, out3 = HeadLossFusedFn.apply(h3, self.decoder[4].weight, self.decoder[4].bias, template.reshape(bs, -1),
                                           gt.reshape(bs, -1), bs, k_rec, k_vel, unit_grad)
        return pred.view(bs, -1, 3), {"loss": out3[0], "rec_loss": out3[1].detach(), "vel_loss": out3[2].detach()}

    store_pred = True     # train_step_direct: False = the fused head writes no prediction (labelled bench variant; the loss and every gradient are unchanged)

    @torch.no_grad()
    def train_step_direct(self, x, one_hot, template, gt, k_rec: float = 1.0, k_vel: float = 10.0, phase: int = 0):
        """Autograd-free training pass for the trainer: forward + VocaLoss + backward as one fixed launch sequence that
        OVERWRITES every parameter's preallocated ``.grad`` (the flat-gradient views of ``FlatAdam``) - no autograd
        graph, no gradient-accumulation kernels, no zero fill, hipGraph-capturable.  Same arithmetic as
        ``forward_with_loss(...)["loss"].backward()``.  Returns (pred (B,V,3), out3 = [loss, rec, vel]).

        ``phase`` splits the sequence for data parallel overlap: 1 = trunk forward + fused head (the gradients of
        ``decoder.4`` - 92 % of all gradient bytes - are final afterwards and can be all-reduced while phase 2 runs),
        2 = trunk backward from the state phase 1 left on the module; 0 = both."""
        be = ops.backend()
        tp = self._trunk_params()
        if phase == 2:
            saved, dH, ret = self._direct_state
            self._direct_state = None
            voca_trunk_bwd(be, saved, dH, out=[p.grad for p in tp])
            return ret
        bs = voca_frames(x)
        if bs % 2 != 0:
            raise RuntimeError(f"VocaLoss pairs consecutive frames (loss.py:33-35): batch {bs} must be even")
        head_w, head_b = self.decoder[4].weight, self.decoder[4].bias
        for p in tp + [head_w, head_b]:
            if p.grad is None or not p.grad.is_contiguous():
                raise RuntimeError("train_step_direct needs preallocated contiguous .grad tensors (see trainer.FlatAdam)")
        h3, saved = voca_trunk_fwd(be, x, one_hot, [p.data for p in tp])
        n, K = head_w.shape
        dev = h3.device
        pred = torch.empty(bs, n, device=dev, dtype=torch.float32) if self.store_pred else None
        out3 = torch.empty(3, device=dev, dtype=torch.float32)
        dH = torch.empty(bs, K, device=dev, dtype=torch.float32)
        be.head_fused(h3, head_w.data, head_b.data, template.reshape(bs, n).contiguous(), bs, gt.reshape(bs, n).contiguous(), bs, bs, n, K,
                      float(k_rec), float(k_vel), pred, out3, dH, head_w.grad, head_b.grad)
        ret = (pred.view(bs, -1, 3) if pred is not None else None, out3)
        if phase == 1:
            self._direct_state = (saved, dH, ret)
            return ret
        voca_trunk_bwd(be, saved, dH, out=[p.grad for p in tp])
        return ret

    def early_grad_params(self):
        """Parameters whose gradients are final after phase 1 of ``train_step_direct``."""
        return [self.decoder[4].weight, self.decoder[4].bias]
