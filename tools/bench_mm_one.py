#!/usr/bin/env python3
"""One a2f_mm launch shape in a loop (PMC passes: tools/pmc_traffic.sh): the Linear-forward layout (NT) in bf16x3 on random data with the
FaceFormer step's FFN-up launch - split-32 image operands (csrc/gemm_img.hip), bias + GELU, the activation written as an image and the saved
derivative as second output (MM_IMG=0: the fp32-operand kernel of rounds 2-3).  usage: bench_mm_one.py M N K [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd  # noqa: F401
from a2f_amd import ops
from a2f_amd.ops import MM_ACT_GELU_SAVE_GRAD, MMOp

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 6
plain = os.environ.get("MM_PLAIN_EPILOGUE") == "1"
be = ops.backend()
dev = "cuda"
x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
y, z = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
img = os.environ.get("MM_IMG", "1") != "0"
if img:
    xi, Wi = torch.empty_like(x), torch.empty_like(W)
    be.split_img(x, M, K, K, xi, K)
    be.split_img(W, N, K, K, Wi, K)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(iters + 2):
    if i == 2:
        e0.record()
    Aop, Wop = (MMOp(xi, K, img=True), MMOp(Wi, K, img=True)) if img else (MMOp(x, K), MMOp(W, K))
    if plain:
        be.mm(Aop, Wop, y, M, N, K, N, precision=1)
    else:
        be.mm(Aop, Wop, y, M, N, K, N, bias=b, act=MM_ACT_GELU_SAVE_GRAD, C2=z, precision=1, c_img=img)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"a2f_mm NT {M}x{N}x{K} {'image operands ' if img else ''}bf16x3 {'plain' if plain else 'bias+GELU+saved derivative'}: {ms:.4f} ms = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s algorithmic")
