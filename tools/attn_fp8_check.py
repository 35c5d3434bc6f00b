#!/usr/bin/env python3
"""fp8 (e4m3) forward mode of the MFMA flash attention (precision 2): error against float64 and time next to the bf16 and bf16x3 modes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import a2f_amd
from a2f_amd import ops
from emu_backend import EmuBackend

be, emu = ops.backend(), EmuBackend()
dev = torch.device("cuda")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for S, T, H in ((2, 600, 12), (1, 1200, 12), (1, 3000, 12), (2, 3000, 12), (1, 3000, 12)):
    ld, ldo = 3 * 64 * H, 64 * H
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(S * T, ld, generator=g)
    ref_ctx, ref_lse = torch.zeros(S * T, ldo), torch.zeros(S, H, T)
    emu.attn_fwd(qkv, S, T, H, ld, 0.125, ref_ctx, ldo, ref_lse)
    row = {"S": S, "T": T}
    for prec, tag in ((1, "bf16x3"), (0, "bf16"), (2, "fp8")):
        ctx, lse = torch.zeros(S * T, ldo, device=dev), torch.zeros(S, H, T, device=dev)
        q = qkv.to(dev)
        be.attn_fwd(q, S, T, H, ld, 0.125, ctx, ldo, lse, precision=prec)
        row[f"{tag}_ctx_rel_l2"] = float((ctx.cpu().double() - ref_ctx.double()).norm() / ref_ctx.double().norm())
        row[f"{tag}_ms"] = round(timeit(lambda: be.attn_fwd(q, S, T, H, ld, 0.125, ctx, ldo, lse, precision=prec)), 4)
    print(json.dumps(row))
