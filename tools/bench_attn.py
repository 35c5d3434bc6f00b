#!/usr/bin/env python3
"""Time the encoder's flash-attention kernels alone (a2f_attn_fwd_img / a2f_attn_bwd_img, train mode: keep bits on the probabilities) at the shapes of the
bench entries: a half-batch of configs[2] (32 x 600) and a half of configs[4] (6 x 3000).  Reports ms and executed bf16 TFLOP/s (x3: bf16x3 products;
forward 4 T^2 64 flops per head, backward 10 T^2 64)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd  # noqa: F401
from a2f_amd import ops

be = ops.backend()
dev = torch.device("cuda")
H, d = 12, 768


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for S, T in ((32, 600), (6, 3000)):
    g = torch.Generator().manual_seed(0)
    M = S * T
    qkv = (torch.randn(M, 3 * d, generator=g) * 0.5).to(dev)
    datt = torch.randn(M, d, generator=g).to(dev)
    att, att_img, lse = torch.empty(M, d, device=dev), torch.empty(M, d, device=dev), torch.empty(S, H, T, device=dev)
    dqkv = torch.empty(M, 3 * d, device=dev)
    TW = be.attn_keep_words(T)
    kq = torch.empty(S * H * T, TW, dtype=torch.int32, device=dev)
    kk = torch.empty(S * H, TW, 32 * TW, dtype=torch.int32, device=dev)
    be.keep_bits_attn(1234, 7, S * H, T, 0.1, kq, kk)
    scale = 0.125
    qkv_i, datt_i = torch.empty_like(qkv), torch.empty_like(datt)
    be.split_img(qkv, M, 3 * d, 3 * d, qkv_i, 3 * d)
    be.split_img(datt, M, d, d, datt_i, d)
    res = {"S": S, "T": T}
    for name, keep_f, keep_b in (("eval", None, None), ("train", (kq, 1.0 / 0.9), (kq, kk, 1.0 / 0.9))):
        for mode, q_in, d_in, flag in (("fp32 in", qkv, datt, False), ("img in", qkv_i, datt_i, True)):
            fwd = timeit(lambda: be.attn_fwd_img(q_in, S, T, H, 3 * d, scale, att, d, lse, att_img, keep=keep_f, qkv_img=flag))
            bwd = timeit(lambda: be.attn_bwd_img(q_in, att, d_in, lse, S, T, H, 3 * d, d, scale, dqkv, keep=keep_b, inputs_img=flag))
            fl = 4.0 * T * T * 64 * S * H
            res[f"{name}, {mode}"] = {"fwd_ms": round(fwd, 4), "bwd_ms": round(bwd, 4), "fwd_PF_executed": round(3 * fl / fwd / 1e12, 3),
                                      "bwd_PF_executed": round(3 * 2.5 * fl / bwd / 1e12, 3)}
    print(json.dumps(res))
