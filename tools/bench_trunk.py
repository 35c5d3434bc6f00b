#!/usr/bin/env python3
"""Micro-benchmark of the VOCA trunk kernels alone at the bench shape (B = 19200 frames): fused forward, fused data
gradients, batched weight gradients.  A2F_LIB_PATH selects an alternative build of the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
from a2f_amd import functional as F, ops

B = int(os.environ.get("HB", 19200)); iters = int(os.environ.get("HI", 20))
dev = "cuda"
torch.manual_seed(0)
be = ops.backend()
model = a2f_amd.Voca(n_verts=15069, n_onehot=8).to(dev)
params = [p.detach() for p in model._trunk_params()]
x = torch.randn(B, 29, 16, device=dev)
oh = torch.zeros(B, 8, device=dev); oh[torch.arange(B), torch.arange(B) % 8] = 1
dh3 = torch.randn(B, 50, device=dev)


def timed(fn):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2] * 1e3


h3, saved = F.voca_trunk_fwd(be, x, oh, params)
print(f"trunk fwd            {timed(lambda: F.voca_trunk_fwd(be, x, oh, params)):8.1f} us")
print(f"trunk bwd (data+wgrad){timed(lambda: F.voca_trunk_bwd(be, saved, dh3)):8.1f} us")
a0, a1, a2, a3, cat, z1, z2, fw0, fw1, fw3, cw1, cw2, cw3, cw4 = saved
new = lambda *s: torch.empty(*s, device=dev)
g = [new(B, 128), new(B, 72), new(B, 64), new(B, 2, 64), new(B, 4, 32), new(B, 8, 32)]
print(f"trunk bwd data only  {timed(lambda: be.voca_trunk_bwd_data(dh3, z2, cat, a3, a2, a1, fw3, fw1, fw0, cw4, cw3, cw2, B, *g)):8.1f} us")
