"""VOCA forward, CPU restatement (test infrastructure, see oracle/__init__.py).

Follows /root/reference/src/model/voca.py:38-49 op for op; parameters arrive as
a dict with the reference's state-dict keys (voca.py:19-36).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

_CONV_IDX = (0, 2, 4, 6)          # nn.Sequential slots holding Conv2d (voca.py:19-28)
_FC_IDX = (0, 1, 3, 4)            # Linear slots; slot 2 is Tanh (voca.py:30-36)


def tile_one_hot(one_hot: torch.Tensor, rows_unused: int = 8, width: int = 16) -> torch.Tensor:
    """voca.py:40-41: slice to 8 subjects, ``repeat(1,16).view(bs,1,-1,16)``.

    Because 16 % 8 == 0 the result is embedding[b,0,r,c] = one_hot[b, c % 8] for every
    row r (the rows are NOT constant copies of the one-hot; this quirk is kept).
    """
    bs = one_hot.shape[0]
    return one_hot[:, :8].repeat(1, width).view(bs, 1, -1, width)


def voca_trunk(P: Dict[str, torch.Tensor], x: torch.Tensor, one_hot: torch.Tensor) -> torch.Tensor:
    """Everything up to the 50-wide bottleneck (input of decoder.4). Returns (bs, 50)."""
    bs = x.shape[0]
    oh8 = one_hot[:, :8]
    z = torch.cat((x.unsqueeze(1), tile_one_hot(one_hot)), 2)       # (bs,1,37,16)  voca.py:42-44
    z = z.permute(0, 2, 3, 1)                                        # (bs,37,16,1)  voca.py:45
    for i in _CONV_IDX:                                              # voca.py:46
        z = F.relu(F.conv2d(z, P[f"time_conv.{i}.weight"], P[f"time_conv.{i}.bias"], stride=(2, 1), padding=(1, 0)))
    z = torch.cat([z.reshape(bs, -1), oh8], 1)                       # (bs,72)       voca.py:47
    z = F.linear(z, P["decoder.0.weight"], P["decoder.0.bias"])
    z = F.linear(z, P["decoder.1.weight"], P["decoder.1.bias"])
    z = torch.tanh(z)
    z = F.linear(z, P["decoder.3.weight"], P["decoder.3.bias"])
    return z


def voca_forward(P: Dict[str, torch.Tensor], x: torch.Tensor, one_hot: torch.Tensor, template: torch.Tensor) -> torch.Tensor:
    """x (bs,29,16), one_hot (bs,>=8), template (bs,V,3) -> (bs,V,3).  voca.py:38-49."""
    bs = x.shape[0]
    z = voca_trunk(P, x, one_hot)
    z = F.linear(z, P["decoder.4.weight"], P["decoder.4.bias"])      # 50 -> n_verts  voca.py:48
    return z.view(bs, -1, 3) + template                              # voca.py:49
