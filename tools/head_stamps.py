#!/usr/bin/env python3
"""Read the per-phase cycle totals of the HF_STAMPS diagnostic build of head_fused (A2F_LIB_PATH=build/abl/liba2f_STAMPS.so).
Shares, not absolute times (stamps forbid overlaps the real kernel has)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd  # noqa: F401
from a2f_amd import ops

B, n, K = 19200, 15069, 50
dev = "cuda"
be = ops.backend()
H = torch.randn(B, K, device=dev); W = torch.randn(n, K, device=dev) * 0.1; bias = torch.randn(n, device=dev)
t = torch.randn(B, n, device=dev) * 5; gt = t + torch.randn(B, n, device=dev) * 0.5
pred = torch.empty(B, n, device=dev); out3 = torch.empty(3, device=dev)
dH = torch.empty(B, K, device=dev); dW = torch.empty(n, K, device=dev); db = torch.empty(n, device=dev)
nb = be.lib.a2f_head_fused_ws_bytes(B, n, K)
ws = torch.zeros(nb // 4, dtype=torch.float32, device=dev)
for _ in range(2):
    be.head_fused(H, W, bias, t, B, gt, B, B, n, K, 1.0, 10.0, pred, out3, dH, dW, db, ws=ws)
torch.cuda.synchronize()
nwg = int(os.environ.get("NWG", 240)); nw = int(os.environ.get("NWAVES", 4))
raw = ws.view(torch.int64)[-(nwg * 8 * 8):].view(nwg, 8, 8).cpu()[:, :nw, :6].double()
tot = raw.sum(-1)
names = ["fwd MFMA", "epilogue", "dW (+tile loads)", "transpose + dH (+stores)", "row-block top (barrier, H request)", "row-block bottom (dH fold, H commit)"]
print("mean cycles per wave: %.0f" % tot.mean())
for i, nm in enumerate(names):
    print(f"  {nm:40s} {raw[..., i].mean():12.0f}  {100 * raw[..., i].sum() / tot.sum():5.1f} %")
