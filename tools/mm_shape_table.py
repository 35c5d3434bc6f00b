#!/usr/bin/env python3
"""Per-shape table of the a2f_mm launches of the 64 x 600 FaceFormer step, measured as bench.py's `gemm_family` leg measures them: one stream, no half-batch
pipeline, no weight-gradient lane, HIP events around every launch (the split-K fold included), two forward + backward passes.  Prints layout, shape, launches,
total ms, share of the family's time and the executed share of the 2.5 PFLOP/s bf16 pipe.  usage: mm_shape_table.py [seqs] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
import bench
from a2f_amd import encoder_ops, ops
from a2f_amd.trainer import TrainStep

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda")
be = ops.backend()
torch.manual_seed(1234)
model = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
with torch.no_grad():
    model.vertice_map_r.weight.normal_(0, 0.1); model.vertice_map.weight.normal_(0, 0.001)
model.train()
model.audio_encoder.config.layerdrop = 0.0                                   # every layer in every pass: the table is per launch, not per step
step = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-4)
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
encoder_ops.WGRAD_STREAM, step.pipeline = False, False
for _ in range(2):
    step(audio, one_hot, gt, templ)
be.prof, be.prof_flops, be.prof_shapes = {}, {}, {}
for _ in range(2):
    step._fwd_bwd(audio, one_hot, gt, templ)
torch.cuda.synchronize()
rows = []
for (lay, M, N, K, batch, win, prec, img), ent in be.prof_shapes.items():
    ms = sum(a.elapsed_time(b) for a, b in ent["events"])
    n = len(ent["events"])
    mult = {0: 1.0, 1: 3.0, 2: 6.0}[prec]
    rows.append((ms, lay, M, N, K, batch, win, prec, img, n, mult * ent["flops"] * n / (ms * 1e-3) / 1e12 / 2500.0))
tot = sum(r[0] for r in rows)
fl = sum(r[10] * r[0] for r in rows) / tot
print(f"a2f_mm launches of 2 x (forward + backward), {S} x {T}: {tot:.1f} ms, time-weighted executed share of the pipe {fl:.3f}")
print(f"{'layout':6s} {'M':>7s} {'N':>6s} {'K':>7s} {'batch':>5s} {'win':>3s} {'prec':>4s} {'img':>3s} {'n':>4s} {'ms':>9s} {'%':>6s} {'pipe':>6s}")
for ms, lay, M, N, K, batch, win, prec, img, n, frac in sorted(rows, reverse=True):
    print(f"{lay:6s} {M:7d} {N:6d} {K:7d} {batch:5d} {int(win):3d} {prec:4d} {int(img):3d} {n:4d} {ms:9.2f} {100 * ms / tot:6.1f} {frac:6.3f}")
