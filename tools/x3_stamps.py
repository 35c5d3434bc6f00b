#!/usr/bin/env python3
"""Per-phase s_memtime totals of the HF_STAMPS build of head_x3_kernel (tools/build_variant.sh stamps -DHF_STAMPS;
A2F_LIB_PATH=build/abl/liba2f_stamps.so).  Shares, not absolute times: the stamps serialise what the real kernel overlaps."""
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
for _ in range(3):
    be.head_fused(H, W, bias, t, B, gt, B, B, n, K, 1.0, 10.0, pred, out3, dH, dW, db, ws=ws)
torch.cuda.synchronize()
nwg = 256
raw = ws.view(torch.int64)[-(nwg * 8 * 8):].view(nwg, 8, 8).cpu().double()
tot = raw.sum(-1)
names = ["0 wait at the top barrier", "1 H request + forward products", "2 tile rows arrive + staging stores", "3 staging barrier",
         "4 chunk read-back + epilogue", "5 G split/transpose + dW (+ tile requests)", "6 dH products + partial stores", "7 fold barrier + fold (+ H commit -> next 0)"]
print("mean ticks per wave: %.0f (min %.0f max %.0f)" % (tot.mean(), tot.min(), tot.max()))
for i, nm in enumerate(names):
    print(f"  {nm:48s} {raw[..., i].mean():12.0f}  {100 * raw[..., i].sum() / tot.sum():5.1f} %")
print("per wave index (share of phase 2 = waiting for the tile rows):", [round(float(raw[:, w, 2].sum() / tot[:, w].sum()), 3) for w in range(8)])
