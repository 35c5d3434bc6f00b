#!/usr/bin/env python3
"""a2f_mm on products of FEW rows (the reference's own batch-1 FaceFormer step: 600 rows through the 12 Transformer layers): time per launch (fold included) of
the Linear-forward (NT) and data-gradient (NN) layouts for a range of split-K factors, fp32 operands, bf16x3.
Round 5 measured with it (profiles/r05_f_small_m_products.txt): 15 - 28 us per launch WHATEVER the split factor (1 .. 16) and whatever the tile (a 64 x 128 tile
configuration built for the purpose gave the same times as 128 x 128 and was dropped) - at 600 rows a product is a launch-latency floor plus ~0.5 us per k-tile,
so the batch-1 step's ~90 such launches are bought back by fewer launches, not by tiling.  usage: bench_mm_small.py [M]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd  # noqa: F401
from a2f_amd import ops
from a2f_amd.ops import MMOp

M = int(sys.argv[1]) if len(sys.argv) > 1 else 600
be = ops.backend()
dev = "cuda"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    Wt = W.t().contiguous()
    y = torch.empty(M, N, device=dev)
    row = {"M": M, "N": N, "K": K}
    for s in (1, 2, 4, 8, 16):
        if K // s < 96:
            continue
        row[f"NT split {s}"] = round(timeit(lambda: be.mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, bias=b, precision=1, split_k=s)), 1)
        row[f"NN split {s}"] = round(timeit(lambda: be.mm(MMOp(x, K), MMOp(Wt, N, trans=True), y, M, N, K, N, precision=1, split_k=s)), 1)
    row["NT auto"] = round(timeit(lambda: be.mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, bias=b, precision=1)), 1)
    print(json.dumps(row), flush=True)
