#!/usr/bin/env python3
"""Microbenchmark of a2f_mm (csrc/gemm_mfma.hip) on the encoder's GEMM shapes, one GPU.  Reports fp32-equivalent TFLOP/s
(2 M N K / time) for bf16x3 (precision 1) and plain bf16 (precision 0), with torch.matmul (rocBLAS fp32 / bf16) beside
it as a yardstick on the same random data.  Interleaved rounds in one process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
from a2f_amd import ops
from a2f_amd.ops import MMOp

dev = torch.device("cuda")
be = ops.backend()


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


def shape(M, N, K, name):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    dW = torch.empty(N, K, device=dev)
    xb, Wb = x.bfloat16(), W.bfloat16()
    res = {"shape": name, "M": M, "N": N, "K": K}
    fl = 2.0 * M * N * K / 1e12
    for prec, tag in ((1, "x3"), (0, "bf16")):
        res[f"fwd_{tag}_TF"] = round(fl / (timeit(lambda: be.mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, precision=prec)) * 1e-3), 1)
        res[f"dgrad_{tag}_TF"] = round(fl / (timeit(lambda: be.mm(MMOp(dy, N), MMOp(W, K, trans=True), dx, M, K, N, K, precision=prec)) * 1e-3), 1)
        res[f"wgrad_{tag}_TF"] = round(fl / (timeit(lambda: be.mm(MMOp(dy, N, trans=True), MMOp(x, K, trans=True), dW, N, K, M, K, split_k=0, precision=prec)) * 1e-3), 1)
    res["torch_f32_fwd_TF"] = round(fl / (timeit(lambda: torch.matmul(x, W.t())) * 1e-3), 1)
    res["torch_f32_wgrad_TF"] = round(fl / (timeit(lambda: torch.matmul(dy.t(), x)) * 1e-3), 1)
    res["torch_bf16_fwd_TF"] = round(fl / (timeit(lambda: torch.matmul(xb, Wb.t())) * 1e-3), 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    shape(9600, 768, 768, "qkv/out projection, 16 x 600 frames")
    shape(9600, 3072, 768, "ffn up")
    shape(9600, 768, 3072, "ffn down")
    shape(16 * 15999, 512, 1536, "feature-encoder conv 1 as implicit GEMM (dense proxy)")
    shape(3000, 768, 768, "one 3000-frame sequence")
