#!/usr/bin/env python3
"""Phase breakdown of the fused trunk kernels from a -DVT_STAMPS build (s_memtime, 100 MHz): wave 0 of workgroup 0.
usage: A2F_LIB_PATH=build/abl/liba2f_vtstamps.so python tools/trunk_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
from a2f_amd import functional as F, ops

B = 19200
dev = "cuda"
be = ops.backend()
model = a2f_amd.Voca(n_verts=15069, n_onehot=8).to(dev)
params = [p.detach() for p in model._trunk_params()]
x = torch.randn(B, 29, 16, device=dev)
oh = torch.zeros(B, 8, device=dev); oh[torch.arange(B), torch.arange(B) % 8] = 1
dh3 = torch.randn(B, 50, device=dev)
h3, saved = F.voca_trunk_fwd(be, x, oh, params)
F.voca_trunk_bwd(be, saved, dh3)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
be.lib.a2f_debug_trunk_stamps.argtypes = [ctypes.c_void_p]
be.lib.a2f_debug_trunk_stamps(buf)        # clear
N = 10
for _ in range(N):
    h3, saved = F.voca_trunk_fwd(be, x, oh, params)
    F.voca_trunk_bwd(be, saved, dh3)
torch.cuda.synchronize()
be.lib.a2f_debug_trunk_stamps(buf)
fw = ["wait top barrier", "stage x -> LDS (+prefetch issue)", "conv1 (after the a0 copy)", "conv2 (+zeroing, one-hot)", "conv3", "conv4", "fc0", "fc1+tanh", "fc3", "a0 copy LDS -> HBM"]
bw = ["wait top barrier", "dh -> LDS + mask loads issue", "fc3' (+tanh')", "fc1'", "fc0'", "conv4'", "conv3'", "conv2'"]
for k, names in ((0, fw), (1, bw)):
    tot = sum(buf[k * 16 + i] for i in range(len(names)))
    print("forward" if k == 0 else "backward-data", "workgroup 0: %.1f us per launch (3 blocks)" % (tot / N / 100.0))
    for i, n in enumerate(names):
        print("   %-34s %7.2f us" % (n, buf[k * 16 + i] / N / 100.0))
