#!/usr/bin/env python3
"""Run-to-run determinism of the VOCA direct training step at the bench size (19 200 frames): N fresh, identically initialised models take one step each (forward +
fused head / loss + backward); every gradient, the prediction and the loss must be BITWISE those of the first.  Fresh models matter: the buffers land at other
addresses each time, which is what exposed the round-4 head-kernel fault (2 of 10 instances differed in 16 head columns; 0 of 400 after the fix).
usage: determinism_probe.py [N = 40]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import a2f_amd  # noqa: E402
from a2f_amd.data import feature_windows  # noqa: E402
from a2f_amd.trainer import TrainStep  # noqa: E402

DEV = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("DIAG_N", "40"))
S, T = 32, 600
g = torch.Generator().manual_seed(21)
feats = torch.randn(S, T, 29, generator=g)
B = S * T
subj = torch.randint(0, 8, (S,), generator=g)
oh = torch.zeros(S, 12); oh[torch.arange(S), subj] = 1.0
oh = oh.repeat_interleave(T, 0).to(DEV)
templ = (torch.randn(8, 5023, 3, generator=g) * 5)[subj].repeat_interleave(T, 0).to(DEV)
gg = torch.Generator(device=DEV).manual_seed(22)
gt = templ + torch.randn(B, 5023, 3, generator=gg, device=DEV) * 0.5
xw = feature_windows(feats).to(DEV)


def fresh_step():
    torch.manual_seed(5)
    m = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
    TrainStep(m, a2f_amd.VocaLoss(), lr=1e-4, use_graph=False)          # flat parameter / gradient buffers
    torch.cuda.synchronize()
    pred, out3 = m.train_step_direct(xw, oh, templ, gt, 1.0, 10.0, phase=0)
    torch.cuda.synchronize()
    out = {"pred": pred.reshape(B, -1).clone(), "out3": out3.clone()}
    for k, p in m.named_parameters():
        out[k] = p.grad.detach().clone().reshape(p.shape[0], -1) if p.dim() > 1 else p.grad.detach().clone()
    return out


def where(a, b):
    ne = a != b
    if a.dim() == 1:
        idx = ne.nonzero().flatten()
        return f"{int(ne.sum())} differ, idx {idx[:4].tolist()}..{idx[-2:].tolist()} max {float((a - b).abs().max()):.3e}"
    rows, cols = ne.any(1).nonzero().flatten(), ne.any(0).nonzero().flatten()
    return f"{int(ne.sum())} differ; rows {len(rows)}: {rows[:4].tolist()}..; cols {len(cols)}: {cols[:4].tolist()}..{cols[-2:].tolist()}; max {float((a - b).abs().max()):.3e}"


base, bad = fresh_step(), 0
for i in range(N):
    r = fresh_step()
    diff = [k for k in base if not torch.equal(base[k], r[k])]
    if diff:
        bad += 1
        print(f"instance {i}: " + " | ".join(f"{k}: {where(base[k], r[k])}" for k in diff), flush=True)
print(f"{bad} of {N} fresh instances differ from the first")
