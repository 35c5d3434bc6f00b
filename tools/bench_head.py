#!/usr/bin/env python3
"""Micro-benchmark of the fused head kernel alone (bench shape: B = 19200 frames, n = 15069, K = 50)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd  # noqa: F401
from a2f_amd import ops

B = int(os.environ.get("HB", 19200)); n = int(os.environ.get("HN", 15069)); K = int(os.environ.get("HK", 50)); iters = int(os.environ.get("HI", 20))
dev = "cuda"
be = ops.backend()
H = torch.randn(B, K, device=dev); W = torch.randn(n, K, device=dev) * 0.1; bias = torch.randn(n, device=dev)
t = torch.randn(B, n, device=dev) * 5; gt = t + torch.randn(B, n, device=dev) * 0.5
pred = torch.empty(B, n, device=dev); out3 = torch.empty(3, device=dev)
dH = torch.empty(B, K, device=dev); dW = torch.empty(n, K, device=dev); db = torch.empty(n, device=dev)
for _ in range(3):
    be.head_fused(H, W, bias, t, B, gt, B, B, n, K, 1.0, 10.0, pred, out3, dH, dW, db)
torch.cuda.synchronize()
be.prof = {}
for _ in range(iters):
    be.head_fused(H, W, bias, t, B, gt, B, B, n, K, 1.0, 10.0, pred, out3, dH, dW, db)
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in be.prof["head_fused"]]
ms.sort()
byts = B * (3 * n * 4 + 2 * K * 4)
be.profile_head(True)
for _ in range(iters):
    be.head_fused(H, W, bias, t, B, gt, B, B, n, K, 1.0, 10.0, pred, out3, dH, dW, db)
torch.cuda.synchronize()
kms, nl = be.profile_head_ms()
be.profile_head(False)
print(f"head kernel alone (in-library HIP events, {nl} launches): {kms:.4f} ms = {B * (3 * n * 4 + 2 * K * 4) / kms / 1e6:.0f} GB/s algorithmic")
print(f"head_fused B={B} K={K}: median {ms[len(ms)//2]:.4f} ms  min {ms[0]:.4f} ms  -> {byts / ms[len(ms)//2] / 1e6:.1f} GB/s algorithmic (incl. reduce kernels)")
