#!/usr/bin/env python3
"""Diagnostic: every fp32 tensor that torch.empty / empty_like hands out on the GPU arrives filled with NaN, then one pipelined FaceFormer train step runs.
A kernel that consumes an element nobody wrote shows as NaN in the loss or in a parameter gradient (named below).  usage: poison_probe.py [seqs] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def _poisoned(t):
    if t.is_cuda and t.dtype == torch.float32 and t.numel() > 0:
        t.fill_(float("nan"))
    elif t.is_cuda and t.dtype == torch.int32 and t.numel() > 0:
        t.fill_(0x55555555)
    return t


torch.empty = lambda *a, **k: _poisoned(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poisoned(_empty_like(*a, **k))

import a2f_amd
import bench
from a2f_amd.trainer import TrainStep

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device("cuda")
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hits = 0
for rep in range(N):
    torch.manual_seed(1234)
    m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
    with torch.no_grad():
        m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
    m.train()
    step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
    step.pipeline_min_seqs, step.pipeline_min_frames = 0, 1
    torch.manual_seed(99)
    for it in range(2):
        out = step(audio, one_hot, gt, templ)
        torch.cuda.synchronize()
        g = step.opt.flat_g
        bad = [name for name, p in m.named_parameters() if id(p) in step.opt._offsets
               and not torch.isfinite(g[step.opt._offsets[id(p)][0]: step.opt._offsets[id(p)][0] + p.numel()]).all()]
        if bad or not bool(torch.isfinite(out["loss"])) or N == 1:
            print(f"rep {rep} step {it}: loss {float(out['loss'].detach()):.6f}; gradients with NaN: {len(bad)}", bad[:60], flush=True)
        if bad:
            hits += 1
    step.close()
    del step, m
print("repetitions with a NaN gradient:", hits, "of", N)
