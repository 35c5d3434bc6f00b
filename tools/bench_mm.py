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


def attention(S, T, H=12):
    """a2f_attn_fwd / a2f_attn_bwd at the encoder's shape: (TFLOP/s of 4 S H T^2 64 forward, 10 S H T^2 64 backward algorithmic flops)."""
    ld, ldo = 3 * 64 * H, 64 * H
    qkv = torch.randn(S * T, ld, device=dev)
    dctx = torch.randn(S * T, ldo, device=dev)
    ctx, lse, dqkv = torch.empty(S * T, ldo, device=dev), torch.empty(S, H, T, device=dev), torch.empty(S * T, ld, device=dev)
    res = {"shape": f"attention S={S} T={T} H={H}"}
    for prec, tag in ((1, "x3"), (0, "bf16")):
        tf = timeit(lambda: be.attn_fwd(qkv, S, T, H, ld, 0.125, ctx, ldo, lse, precision=prec))
        tb = timeit(lambda: be.attn_bwd(qkv, ctx, dctx, lse, S, T, H, ld, ldo, 0.125, dqkv, precision=prec))
        res[f"fwd_{tag}_ms"], res[f"bwd_{tag}_ms"] = round(tf, 4), round(tb, 4)
        res[f"fwd_{tag}_TF"] = round(4.0 * S * H * T * T * 64 / (tf * 1e-3) / 1e12, 1)
        res[f"bwd_{tag}_TF"] = round(10.0 * S * H * T * T * 64 / (tb * 1e-3) / 1e12, 1)
    q = qkv.view(S, T, 3, H, 64).permute(2, 0, 3, 1, 4).contiguous()
    res["torch_sdpa_f32_fwd_ms"] = round(timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q[0], q[1], q[2])), 4)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if os.environ.get("MM_QUICK") == "1":                 # PMC / ablation runs: two GEMM shapes and the attention kernels only
        shape(9600, 768, 768, "qkv/out projection, 16 x 600 frames")
        shape(9600, 768, 3072, "ffn down")
        attention(16, 600)
        sys.exit(0)
    attention(16, 600)
    attention(2, 3000)
    shape(9600, 768, 768, "qkv/out projection, 16 x 600 frames")
    shape(9600, 3072, 768, "ffn up")
    shape(9600, 768, 3072, "ffn down")
    shape(16 * 15999, 512, 1536, "feature-encoder conv 1 as implicit GEMM (dense proxy)")
    shape(3000, 768, 768, "one 3000-frame sequence")
