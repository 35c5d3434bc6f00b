#!/usr/bin/env python3
"""How far is ONE dropout mask per frame row (what a2f_amd's incremental decoder draws) from the reference's fresh masks at every step of
its O(T^3) loop?  BUILD CONTAINER ONLY (imports /root/reference).  Runs the reference FaceFormer (pinned encoder output, formula weights) in
train mode N times each way - (a) untouched: torch redraws every nn.Dropout / attention-probability mask at every loop step;
(b) with the dropout calls replaced by per-row masks (the G14 patch, fresh masks per run) - and compares the two output distributions
per vertex coordinate: mean and standard deviation over the runs, against the eval-mode output as the scale."""
import math, os, sys
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
import gen_golden as G                                   # build_reference_faceformer, decoder_state, FF_GAINS
from oracle import weights as W

T, N = int(os.environ.get("DD_T", "24")), int(os.environ.get("DD_N", "200"))
torch.manual_seed(0)
m = G.build_reference_faceformer(stub_encoder=True)
m.load_state_dict(G.decoder_state(m, G.FF_GAINS), strict=False)
enc = W.formula_input("dd.enc", (1, T, 768))
m.audio_encoder.enc = enc
audio = torch.zeros(1, T * 16000 // 60 + 8)
one_hot = torch.zeros(1, 12); one_hot[0, 3] = 1.0
template = torch.zeros(1, 5023, 3)
sel = torch.arange(0, 15069, 37)                         # a sample of vertex coordinates

m.eval()
with torch.no_grad():
    base = m(audio, one_hot, template).reshape(T, -1)[:, sel]

real_dropout, real_sdpa = F.dropout, F.scaled_dot_product_attention


def run_per_step():
    with torch.no_grad():
        return m(audio, one_hot, template).reshape(T, -1)[:, sel]


def run_per_row(gen):
    draw = lambda *shape: torch.rand(*shape, generator=gen) >= 0.1
    masks = {"ppe": draw(T, 64), "d1": draw(T, 64), "d2": draw(T, 64), "act": draw(T, 128), "d3": draw(T, 64), "attn": draw(4, T, T), "cross": draw(4, T, T)}
    order, calls = ("ppe", "d1", "d2", "act", "d3"), {"drop": 0, "sdpa": 0}

    def row_dropout(x, p=0.5, training=True, inplace=False):
        site = order[calls["drop"] % 5]; calls["drop"] += 1
        return x * masks[site][: x.shape[1]].to(x.dtype) / 0.9

    def row_sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, **kw):
        site = ("attn", "cross")[calls["sdpa"] % 2]; calls["sdpa"] += 1
        n, nk = q.shape[-2], k.shape[-2]
        pr = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1]) + attn_mask, dim=-1)
        return torch.matmul(pr * masks[site][:, :n, :nk].to(q.dtype).view(1, 4, n, nk) / 0.9, v)

    F.dropout, F.scaled_dot_product_attention = row_dropout, row_sdpa
    try:
        with torch.no_grad():
            return m(audio, one_hot, template).reshape(T, -1)[:, sel]
    finally:
        F.dropout, F.scaled_dot_product_attention = real_dropout, real_sdpa


m.train()
gen = torch.Generator().manual_seed(1)
A = torch.stack([run_per_step() for _ in range(N)])      # (N, T, sel)
B = torch.stack([run_per_row(gen) for _ in range(N)])
scale = base.abs().mean()
rep = {
    "T": T, "runs": N,
    "eval_output_mean_abs": float(scale),
    "mean_difference_per_step_vs_per_row_rel": float((A.mean(0) - B.mean(0)).abs().mean() / scale),
    "standard_error_of_that_difference_rel": float(((A.var(0) + B.var(0)) / N).sqrt().mean() / scale),
    "std_per_step_rel": float(A.std(0).mean() / scale),
    "std_per_row_rel": float(B.std(0).mean() / scale),
    "train_mean_vs_eval_per_step_rel": float((A.mean(0) - base).abs().mean() / scale),
    "train_mean_vs_eval_per_row_rel": float((B.mean(0) - base).abs().mean() / scale),
}
import json
print(json.dumps(rep, indent=1))
