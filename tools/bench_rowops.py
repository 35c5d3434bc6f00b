#!/usr/bin/env python3
"""Time the row kernels of a Transformer layer at the half-batch shape of configs[2] (19 200 x 768): LayerNorm forward / backward in their image forms."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd  # noqa: F401
from a2f_amd import ops

be = ops.backend()
dev = torch.device("cuda")
R, C = 19200, 768
g = torch.Generator().manual_seed(0)
x, dy = torch.randn(R, C, generator=g).to(dev), torch.randn(R, C, generator=g).to(dev)
gam, bet = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
y, y_img, mean, rstd = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
dx, dx_img, dg, db, cs = torch.empty(R, C, device=dev), torch.empty(R, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (R, C // 32), device=dev, dtype=torch.int32)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


fwd = timeit(lambda: be.layernorm_fwd_img(x, R, C, gam, bet, 1e-5, y, y_img, mean, rstd))
bwd = timeit(lambda: be.layernorm_bwd_img(dy, x, mean, rstd, gam, R, C, dx, dg, db, dxm_colsum=cs, keep=(bits, 1.0 / 0.9), dxm_img=dx_img))
mb = R * C * 4 / 1e6
print(json.dumps({"rows": R, "cols": C, "layernorm_fwd_img_ms": round(fwd, 4), "fwd_GBs": round(3 * mb / fwd, 1), "layernorm_bwd_img_ms (incl. fold)": round(bwd, 4),
                  "bwd_GBs": round(4 * mb / bwd, 1)}))
