#!/usr/bin/env python3
"""Time the FaceFormer decoder recurrence kernels alone (a2f_ff_decoder_fwd / _bwd, and their dropout variants): S sequences x T frames."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a2f_amd
from a2f_amd import ops

be = ops.backend()
dev = torch.device("cuda")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


nP = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64 + 6 * 64 + 64 * 64 + 64
for S, T in ((32, 600), (64, 600), (12, 3000)):
    g = torch.Generator().manual_seed(0)
    P = (torch.randn(nP, generator=g) * 0.1).to(dev)
    pe = torch.randn(60, 64, generator=g).to(dev)
    cross, style, dy = torch.randn(S, T, 64, generator=g).to(dev), torch.randn(S, 64, generator=g).to(dev), torch.randn(S, T, 64, generator=g).to(dev)
    y = torch.empty(S, T, 64, device=dev)
    save = torch.empty(be.ff_save_floats(S, T), device=dev)
    dP, dc, ds = torch.empty(nP, device=dev), torch.empty(S, T, 64, device=dev), torch.empty(S, 64, device=dev)
    row = torch.empty(S * T, 10, dtype=torch.int32, device=dev); attn = torch.empty(S * 4 * T, be.attn_keep_words(T), dtype=torch.int32, device=dev)
    be.keep_bits(1, 1000, S * T, 320, 0.1, row); be.keep_bits(1, 1001, S * 4 * T, T, 0.1, attn)
    drop = (row, attn, 1 / 0.9)
    r = {"S": S, "T": T}
    r["fwd_ms"] = round(timeit(lambda: be.ff_decoder_fwd(P, pe, cross, style, S, T, 60, y, save)), 3)
    r["bwd_ms"] = round(timeit(lambda: be.ff_decoder_bwd(P, dy, S, T, 60, save, dP, dc, ds)), 3)
    r["fwd_drop_ms"] = round(timeit(lambda: be.ff_decoder_fwd(P, pe, cross, style, S, T, 60, y, save, drop=drop)), 3)
    r["bwd_drop_ms"] = round(timeit(lambda: be.ff_decoder_bwd(P, dy, S, T, 60, save, dP, dc, ds, drop=drop)), 3)
    print(json.dumps(r), flush=True)
