"""Drop-in ``Audio2Mesh`` (reference: /root/reference/src/model/audio2face.py:5-69) on the HIP hot path.

Same constructor, sub-module names / shapes / construction order (state-dict and seeded-init compatible),
``forward(x, one_hot, template, **kwargs)`` / ``predict``.  Convolutions run as implicit MFMA GEMMs over
channels-last activations, BatchNorm (training-mode batch statistics, running-buffer updates) and the ReLUs are
HIP kernels; nothing dispatches to torch.nn at run time.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..functional import A2M_CH, A2MPackBuffers, A2MTrunkMMFn, HeadFn, HeadLossFusedFn


class Audio2Mesh(nn.Module):
    def __init__(self, n_verts: int, n_onehot: int):
        super().__init__()
        self.n_verts = n_verts
        self.n_onehot = n_onehot
        ana = []
        for l in range(5):                                                    # audio2face.py:13-29
            ana += [nn.Conv2d(A2M_CH[l], A2M_CH[l + 1], kernel_size=(1, 3), stride=(1, 2), padding=(0, 1)),
                    nn.BatchNorm2d(A2M_CH[l + 1]), nn.ReLU()]
        self.analysis_net = nn.Sequential(*ana)
        art = []
        for _ in range(3):                                                    # audio2face.py:32-40
            art += [nn.Conv2d(256, 256, kernel_size=(3, 1), stride=(2, 1), padding=(1, 0)), nn.BatchNorm2d(256), nn.ReLU()]
        art += [nn.BatchNorm2d(256), nn.Conv2d(256, 256, kernel_size=(3, 1), stride=(2, 1), padding=(1, 0)), nn.ReLU()]   # :41-43
        art += [nn.BatchNorm2d(256), nn.Conv2d(256, 256, kernel_size=(4, 1), stride=(4, 1)), nn.ReLU()]                   # :44-46
        self.articulation_net = nn.Sequential(*art)
        self.output_net = nn.Sequential(nn.Linear(256 + n_onehot, 72), nn.Linear(72, 128), nn.Tanh(), nn.Linear(128, 50),
                                        nn.Linear(50, n_verts))               # audio2face.py:49-55

    # layer order used by A2MTrunkMMFn: 5 analysis blocks, 3 conv->BN blocks, 2 BN->conv blocks
    def _convs(self):
        return [self.analysis_net[3 * l] for l in range(5)] + [self.articulation_net[i] for i in (0, 3, 6, 10, 13)]

    def _bns(self):
        return [self.analysis_net[3 * l + 1] for l in range(5)] + [self.articulation_net[i] for i in (1, 4, 7, 9, 12)]

    def trunk(self, x, one_hot):
        bns = self._bns()
        bufs = []
        for b in bns:
            bufs += [b.running_mean, b.running_var]
        params = []
        for c in self._convs():
            params += [c.weight, c.bias]
        for b in bns:
            params += [b.weight, b.bias]
        for i in (0, 1, 3):
            params += [self.output_net[i].weight, self.output_net[i].bias]
        # the step's packed / padded parameter copies live in buffers this module owns (one set per device; see functional.A2MPackBuffers)
        table = self.__dict__.setdefault("_a2f_pack", {})
        pk = table.get(x.device)
        if pk is None:
            pk = table[x.device] = A2MPackBuffers(x.device, self.articulation_net[13].weight.shape[2])
        h3 = A2MTrunkMMFn.apply(x, one_hot, self.training, pk, *bufs, *params)
        if self.training:
            torch._foreach_add_([b.num_batches_tracked for b in bns], 1)          # one launch instead of ten
        return h3

    def forward(self, x, one_hot, template, **kwargs):
        bs = x.size(0)
        h3 = self.trunk(x, one_hot)
        pred = HeadFn.apply(h3, self.output_net[4].weight, self.output_net[4].bias, template.reshape(bs, -1))
        return pred.view(bs, -1, 3)

    def predict(self, x, one_hot, template, **kwargs):
        return self(x, one_hot, template, **kwargs)

    def forward_with_loss(self, x, one_hot, template, gt, k_rec: float = 1.0, k_vel: float = 10.0, unit_grad: bool = False):
        """Training fast path (same contract as ``Voca.forward_with_loss``): ``output_net.4`` + template add + VocaLoss +
        their backward in one pass over the (B, 15069) frame matrix instead of a GEMM, two loss passes and two more
        GEMMs.  Returns (pred (B,V,3) detached, {"loss","rec_loss","vel_loss"}); only "loss" carries gradient."""
        bs = x.size(0)
        if bs % 2 != 0:
            raise RuntimeError(f"VocaLoss pairs consecutive frames (loss.py:33-35): batch {bs} must be even")
        h3 = self.trunk(x, one_hot)
        pred, out3 = HeadLossFusedFn.apply(h3, self.output_net[4].weight, self.output_net[4].bias, template.reshape(bs, -1),
                                           gt.reshape(bs, -1), bs, k_rec, k_vel, unit_grad)
        return pred.view(bs, -1, 3), {"loss": out3[0], "rec_loss": out3[1].detach(), "vel_loss": out3[2].detach()}
