#!/usr/bin/env python3
"""How long the HOST needs to issue one FaceFormer train step (time until TrainStep returns, nothing synchronised) next to the
step time on the GPU: if the two are close the step is launch-bound on the host, not on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a2f_amd
from a2f_amd.trainer import TrainStep
import bench

S, T = int(os.environ.get("FF_S", "64")), 600
dev = torch.device("cuda")
torch.manual_seed(1234)
m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
with torch.no_grad():
    m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
m.train()
step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
for _ in range(3):
    step(audio, one_hot, gt, templ)
torch.cuda.synchronize()
iss, tot = [], []
for _ in range(6):
    t0 = time.perf_counter()
    step(audio, one_hot, gt, templ)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"S={S}: host issue {sum(iss)/len(iss):.1f} ms per step, step (issue + drain) {sum(tot)/len(tot):.1f} ms")
