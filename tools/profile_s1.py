#!/usr/bin/env python3
"""Where does a one-sequence FaceFormer train step (the reference's batch = 1) spend its HOST time?  cProfile over 5 steps at S = 1, T = 600."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
from a2f_amd.trainer import TrainStep

dev = torch.device("cuda")
S, T = int(os.environ.get("S", 1)), 600
torch.manual_seed(0)
m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
with torch.no_grad():
    m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
m.train()
step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
audio = torch.randn(S, T * 16000 // 60 + 8, device=dev) * 0.1
oh = torch.zeros(S, 12, device=dev); oh[:, 3] = 1
templ = torch.randn(S, 5023, 3, device=dev) * 5
gt = templ.view(S, 1, 5023, 3) + torch.randn(S, T, 5023, 3, device=dev) * 0.5
for _ in range(3):
    step(audio, oh, gt, templ)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step(audio, oh, gt, templ)
t1 = time.perf_counter()            # host issue time (no sync)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"S={S}: host issue {1e3 * (t1 - t0) / 5:.2f} ms/step, wall {1e3 * (t2 - t0) / 5:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(audio, oh, gt, templ)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
