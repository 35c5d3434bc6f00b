#!/usr/bin/env python3
"""Is a FaceFormer step bound by the host's launch rate?  Per step: the time the Python call takes to RETURN (everything enqueued) against the time until the GPU has
finished; then a cProfile of the enqueue side.  usage: host_bound_probe.py [S] [T]   (default 1 600: the reference's own batch-1 step, train.py:33)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a2f_amd
from a2f_amd.trainer import TrainStep
from bench import make_ff_batch

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda")
torch.manual_seed(1234)
model = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
with torch.no_grad():
    model.vertice_map_r.weight.normal_(0, 0.1)
    model.vertice_map.weight.normal_(0, 0.001)
model.train()
step = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-4)
audio, one_hot, gt, templ = make_ff_batch(S, T, 0, dev)
for _ in range(5):
    step(audio, one_hot, gt, templ)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(20):
    t0 = time.perf_counter()
    step(audio, one_hot, gt, templ)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
enq.sort(); tot.sort()
print(f"S={S} T={T}: host enqueue {enq[len(enq) // 2]:.2f} ms per step (median), step with a synchronisation after it {tot[len(tot) // 2]:.2f} ms")
# back-to-back steps (what the bench times): the host runs ahead of the GPU if it can
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step(audio, one_hot, gt, templ)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"20 steps back to back: host done after {1e3 * (t1 - t0) / 20:.2f} ms per step, GPU after {1e3 * (t2 - t0) / 20:.2f} ms per step")
torch.autograd.set_multithreading_enabled(False)          # the backward's Python (Function.backward) in this thread, where the profiler sees it
for _ in range(3):
    step(audio, one_hot, gt, templ)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(audio, one_hot, gt, templ)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
