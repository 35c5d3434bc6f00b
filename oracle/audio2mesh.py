"""Audio2Mesh (Karras'17-style CNN) forward, CPU restatement (test infrastructure).

Follows /root/reference/src/model/audio2face.py:57-66; layer tables restate
audio2face.py:13-55.  ``P`` carries the reference's state-dict keys, including
BatchNorm running buffers, which are updated in place in training mode exactly
like ``nn.BatchNorm2d`` (momentum 0.1, eps 1e-5, unbiased running variance).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

# (conv slot, bn slot) in analysis_net: Conv(1,3)/s(1,2)/p(0,1) -> BN -> ReLU   (audio2face.py:13-29)
ANALYSIS = tuple((3 * l, 3 * l + 1) for l in range(5))
# articulation_net (audio2face.py:31-47): three conv->BN->ReLU, then two BN->conv->ReLU (order flip)
ARTIC_POST = ((0, 1), (3, 4), (6, 7))          # conv slot, bn slot (BN after conv)
ARTIC_PRE = ((9, 10), (12, 13))                # bn slot, conv slot (BN before conv)


def _bn(P, key, z, training, momentum=0.1, eps=1e-5):
    return F.batch_norm(z, P[key + ".running_mean"], P[key + ".running_var"], P[key + ".weight"], P[key + ".bias"],
                        training, momentum, eps)


def tile_one_hot(one_hot: torch.Tensor, width: int = 32) -> torch.Tensor:
    """audio2face.py:59: ``one_hot.repeat(1,32).view(bs,1,-1,32)`` -> (bs,1,n_onehot,32);
    element [r,c] = one_hot[(32 r + c) % n_onehot] (rotated copies, quirk kept)."""
    bs = one_hot.shape[0]
    return one_hot.repeat(1, width).view(bs, 1, -1, width)


def audio2mesh_trunk(P: Dict[str, torch.Tensor], x: torch.Tensor, one_hot: torch.Tensor, training: bool) -> torch.Tensor:
    """Up to the 50-wide bottleneck (input of output_net.4). Returns (bs, 50)."""
    bs = x.shape[0]
    z = torch.cat((x.unsqueeze(1), tile_one_hot(one_hot)), 2)                        # (bs,1,64,32)  :60-62
    for c, b in ANALYSIS:
        z = F.conv2d(z, P[f"analysis_net.{c}.weight"], P[f"analysis_net.{c}.bias"], stride=(1, 2), padding=(0, 1))
        z = F.relu(_bn(P, f"analysis_net.{b}", z, training))
    for c, b in ARTIC_POST:                                                          # :32-40
        z = F.conv2d(z, P[f"articulation_net.{c}.weight"], P[f"articulation_net.{c}.bias"], stride=(2, 1), padding=(1, 0))
        z = F.relu(_bn(P, f"articulation_net.{b}", z, training))
    b, c = ARTIC_PRE[0]                                                              # :41-43
    z = _bn(P, f"articulation_net.{b}", z, training)
    z = F.relu(F.conv2d(z, P[f"articulation_net.{c}.weight"], P[f"articulation_net.{c}.bias"], stride=(2, 1), padding=(1, 0)))
    b, c = ARTIC_PRE[1]                                                              # :44-46
    z = _bn(P, f"articulation_net.{b}", z, training)
    z = F.relu(F.conv2d(z, P[f"articulation_net.{c}.weight"], P[f"articulation_net.{c}.bias"], stride=(4, 1)))
    z = torch.cat((z.reshape(bs, -1), one_hot), 1)                                   # (bs,268)  :64-65
    z = F.linear(z, P["output_net.0.weight"], P["output_net.0.bias"])
    z = F.linear(z, P["output_net.1.weight"], P["output_net.1.bias"])
    z = torch.tanh(z)
    z = F.linear(z, P["output_net.3.weight"], P["output_net.3.bias"])
    return z


def audio2mesh_forward(P, x, one_hot, template, training: bool = False) -> torch.Tensor:
    """x (bs,52,32), one_hot (bs,12), template (bs,V,3) -> (bs,V,3).  audio2face.py:57-66."""
    bs = x.shape[0]
    z = audio2mesh_trunk(P, x, one_hot, training)
    z = F.linear(z, P["output_net.4.weight"], P["output_net.4.bias"])
    return z.view(bs, -1, 3) + template
