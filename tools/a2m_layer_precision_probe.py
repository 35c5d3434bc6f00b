#!/usr/bin/env python3
"""Which of Audio2Mesh's forward convolutions need the fp32-class (bf16x6) product?  For several per-layer settings of the forward precision: the worst gradient
error against the reference fixture G2 (B = 4, train mode: batch statistics) and - at B = 1024 - the worst gradient distance from the all-bf16x6 step, plus the step time."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import a2f_amd
from a2f_amd import functional as Fn
from golden_util import load, packed_error
from oracle import weights as W          # formula weights only (test infrastructure)

DEV, NV = torch.device("cuda"), 15069
BN_CANCELLED = {f"analysis_net.{3 * l}.bias" for l in range(5)} | {f"articulation_net.{i}.bias" for i in (0, 3, 6)}


def g2_errors():
    g = load("g2_audio2mesh")
    m = a2f_amd.Audio2Mesh(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.audio2mesh_shapes(NV, 12), gains={"output_net.4": 4.0}))
    m.to(DEV).train()
    B = 4
    x = W.formula_input("g2.x", (B, 52, 32)).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g2.template", (B, 5023, 3), 5.0)
    gt = (template + W.formula_input("g2.gt", (B, 5023, 3), 0.5)).to(DEV)
    pred = m(x, one_hot, template.to(DEV))
    a2f_amd.VocaLoss()(pred, gt)["loss"].backward()
    errs = {k: packed_error(g, f"train::grad::{k}", p.grad) for k, p in m.named_parameters() if k not in BN_CANCELLED}
    worst = max(errs, key=errs.get)
    return packed_error(g, "train::pred", pred), errs[worst], worst


def big_step(B=1024):
    torch.manual_seed(0)
    m = a2f_amd.Audio2Mesh(NV, 12).to(DEV).train()
    x = torch.randn(B, 52, 32, device=DEV)
    oh = torch.zeros(B, 12, device=DEV); oh[torch.arange(B), torch.arange(B) % 12] = 1
    t = torch.randn(B, 5023, 3, device=DEV) * 5
    gt = t + torch.randn(B, 5023, 3, device=DEV) * 0.5

    def run():
        for p in m.parameters():
            p.grad = None
        _pred, l = m.forward_with_loss(x, oh, t, gt)
        l["loss"].backward()
    run(); run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    return ms, {k: p.grad.double().clone() for k, p in m.named_parameters()}


settings = {"all bf16x6": {}, "l1 x3": {1: 1}, "l1-2 x3": {1: 1, 2: 1}, "l1-4 x3 (analysis)": {1: 1, 2: 1, 3: 1, 4: 1}, "l5-8 x3 (articulation)": {5: 1, 6: 1, 7: 1, 8: 1},
            "l3-4 x3": {3: 1, 4: 1}, "all x3": {l: 1 for l in range(1, 9)}}
ref = None
for name, layers in settings.items():
    Fn.A2M_PREC_FWD_LAYER.clear()
    Fn.A2M_PREC_FWD_LAYER.update(layers)
    perr, gerr, gk = g2_errors()
    ms, grads = big_step()
    if ref is None:
        ref = grads
    dist = max(float((grads[k] - ref[k]).norm() / (ref[k].norm() + 1e-300)) for k in grads if k not in BN_CANCELLED)
    print(json.dumps({"setting": name, "G2 pred err": float(f"{perr:.2e}"), "G2 worst grad err": float(f"{gerr:.2e}"), "G2 worst grad": gk,
                      "B=1024 eager fwd+bwd ms": round(ms, 3), "B=1024 worst grad distance from all-x6": float(f"{dist:.2e}")}), flush=True)
