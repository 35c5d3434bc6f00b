#!/usr/bin/env python3
"""Microbenchmark of the image-operand a2f_mm kernels (csrc/gemm_img.hip) against the fp32-operand kernels (csrc/gemm_mfma.hip) on the Transformer
shapes of the 64 x 600 FaceFormer step (M = 38 400 rows).  Interleaved rounds in one process, random data; reports ms, fp32-equivalent TFLOP/s
(2 M N K / t) and EXECUTED bf16 TFLOP/s (x3) with its share of the 2.5 PFLOP/s dense bf16 pipe."""
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
M = int(os.environ.get("MM_M", "38400"))
ROUNDS = int(os.environ.get("MM_ROUNDS", "5"))
ITERS = int(os.environ.get("MM_ITERS", "10"))


def time_all(fns):
    """fns: dict name -> callable.  ROUNDS interleaved rounds of ITERS launches each; returns name -> (median ms, min ms)."""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / ITERS)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def img(x):
    R, C = x.shape
    o = torch.empty_like(x)
    be.split_img(x, R, C, C, o, C)
    return o


def shape(N, K, name):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    dy = torch.randn(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev)
    y, y2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    dW = torch.empty(N, K, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (M, (N + 31) // 32), device=dev, dtype=torch.int32)
    xi, Wi, dyi = img(x), img(W), img(dy)
    fl = 2.0 * M * N * K / 1e12
    fns = {
        "fp32 fwd plain": lambda: be.mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, precision=1),
        "img  fwd plain": lambda: be.mm(MMOp(xi, K, img=True), MMOp(Wi, K, img=True), y, M, N, K, N, precision=1),
        "img  fwd plain->img": lambda: be.mm(MMOp(xi, K, img=True), MMOp(Wi, K, img=True), y, M, N, K, N, precision=1, c_img=True),
        "fp32 fwd bias+res+keep": lambda: be.mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, precision=1, bias=bias, aux=aux, ldaux=N, aux_mode=ops.MM_AUX_ADD, keep=(bits, 1.1)),
        "img  fwd bias+res+keep": lambda: be.mm(MMOp(xi, K, img=True), MMOp(Wi, K, img=True), y, M, N, K, N, precision=1, bias=bias, aux=aux, ldaux=N, aux_mode=ops.MM_AUX_ADD,
                                                keep=(bits, 1.1)),
        "fp32 fwd bias+gelu4+keep": lambda: be.mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, precision=1, bias=bias, act=ops.MM_ACT_GELU_SAVE_GRAD, C2=y2, keep=(bits, 1.1)),
        "img  fwd bias+gelu4+keep->img": lambda: be.mm(MMOp(xi, K, img=True), MMOp(Wi, K, img=True), y, M, N, K, N, precision=1, bias=bias, act=ops.MM_ACT_GELU_SAVE_GRAD, C2=y2,
                                                       keep=(bits, 1.1), c_img=True),
        "fp32 wgrad": lambda: be.mm(MMOp(dy, N, trans=True), MMOp(x, K, trans=True), dW, N, K, M, K, split_k=0, precision=1),
        "img  wgrad": lambda: be.mm(MMOp(dyi, N, trans=True, img=True), MMOp(xi, K, trans=True, img=True), dW, N, K, M, K, split_k=0, precision=1),
        "split pass (M x K)": lambda: be.split_img(x, M, K, K, xi, K),
    }
    only = os.environ.get("MM_ONLY")
    if only:
        fns = {k: v for k, v in fns.items() if only in k}
    r = time_all(fns)
    out = {"shape": name, "M": M, "N": N, "K": K, "gflop": round(fl * 1e3, 1)}
    for k, (med, mn) in r.items():
        if k.startswith("split"):
            out[k] = {"ms": round(med, 4), "GB/s": round(8.0 * M * K / med / 1e6, 0)}
        else:
            out[k] = {"ms": round(med, 4), "min_ms": round(mn, 4), "TF_fp32eq": round(fl / med * 1e3, 1), "TF_exec": round(3 * fl / med * 1e3, 0),
                      "pipe_frac": round(3 * fl / med * 1e3 / 2500.0, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    shape(768, 768, "out projection / dgrad Wo")
    shape(2304, 768, "qkv projection")
    shape(3072, 768, "ffn up")
    shape(768, 3072, "ffn down")
