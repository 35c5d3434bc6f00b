#!/usr/bin/env python3
"""Audio2Mesh: gradients of the a2f_mm trunk (bf16x3) against the legacy trunk (bf16x6, fp32-class) and against the exact-fp32
legacy trunk on the same seeded batch, for several batch sizes (the BatchNorm backward's conditioning depends on the rows per
statistic).  Prints the worst relative L2 difference over all parameters."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
from a2f_amd import functional as Fn

dev = torch.device("cuda")


def grads(B, use_mm, prec=2):
    torch.manual_seed(0)
    m = a2f_amd.Audio2Mesh(15069, 12).to(dev).train()
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 52, 32, generator=g).to(dev)
    oh = torch.zeros(B, 12); oh[torch.arange(B), torch.arange(B) % 12] = 1
    t = (torch.randn(B, 5023, 3, generator=g) * 5).to(dev)
    gt = t + (torch.randn(B, 5023, 3, generator=g) * 0.5).to(dev)
    import a2f_amd.model.audio2face as A
    A.A2M_USE_MM = use_mm
    Fn.A2M_GEMM_PRECISION = prec
    pred = m(x, oh.to(dev), t)
    loss = a2f_amd.VocaLoss()(pred, gt)
    loss["loss"].backward()
    return {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}, pred.detach().double().cpu()


for B in (4, 32, 128, 1024):
    ref, pref = grads(B, False, 0)                       # exact fp32 MFMA
    out = {"B": B}
    for name, (use_mm, prec) in {"legacy_bf16x6": (False, 2), "legacy_bf16x3": (False, 1), "mm_bf16x3": (True, 2)}.items():
        g, p = grads(B, use_mm, prec)
        worst, wk = 0.0, None
        scale = max(float(v.norm()) for v in ref.values())
        for k in ref:
            if float(ref[k].norm()) < 1e-5 * scale:          # conv biases in front of a batch-statistics BatchNorm: the gradient is identically zero
                continue
            e = float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30))
            if e > worst:
                worst, wk = e, k
        out[name] = {"worst_grad_rel": round(worst, 7), "param": wk, "pred_rel": round(float((p - pref).norm() / pref.norm()), 9)}
    print(json.dumps(out), flush=True)
