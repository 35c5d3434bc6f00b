#!/usr/bin/env python3
"""Which parameters differ between two runs of the same pipelined FaceFormer train step from the same state and seed (debugging aid for
test_faceformer_full_size_pipelined_step_is_bitwise_reproducible): prints parameter names whose values or gradients are not bitwise equal."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
import bench
from a2f_amd.trainer import TrainStep

S, T = int(os.environ.get("FF_S", 48)), int(os.environ.get("FF_T", 600))
NSTEP = int(os.environ.get("FF_STEPS", 1))
dev = torch.device("cuda")
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
runs = []
for _run in range(2):
    torch.manual_seed(1234)
    m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
    with torch.no_grad():
        m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
    m.train()
    step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
    step.pipeline_min_seqs, step.pipeline_min_frames = 1, 1
    if os.environ.get("FF_NOPIPE") == "1":
        step.pipeline = False
    torch.manual_seed(99)
    for _ in range(NSTEP):
        out = step(audio, one_hot, gt, templ)
    torch.cuda.synchronize()
    runs.append(({k: p.detach().clone() for k, p in m.named_parameters()}, {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                 float(out["loss"])))
    del step, m
print("loss", runs[0][2], runs[1][2])
for what, idx in (("param", 0), ("grad", 1)):
    bad = [k for k in runs[0][idx] if not torch.equal(runs[0][idx][k], runs[1][idx][k])]
    print(what, "differs in", len(bad), "of", len(runs[0][idx]), ":", bad[:4])
    for k in runs[0][idx]:
        if not k.startswith("audio_encoder"):
            a, b = runs[0][idx][k], runs[1][idx][k]
            print("   %-60s equal=%s  max|diff|=%.3e  max|a|=%.3e" % (k, bool(torch.equal(a, b)), float((a - b).abs().max()), float(a.abs().max())))
