#!/usr/bin/env python3
"""Microbenchmark of the feature-encoder convolutions as they run in the model (windowed implicit GEMMs on a2f_mm): forward (+ GELU
epilogue), weight gradient, data gradient per layer shape, fp32-equivalent TFLOP/s, beside the same products on plain (dense) operands."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a2f_amd
from a2f_amd import ops
from a2f_amd.ops import MMOp, MM_ACT_GELU, MM_ACT_GELU_SAVE_GRAD

dev = torch.device("cuda")
be = ops.backend()


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


S = int(os.environ.get("CONV_S", "16"))
for name, Lin, KH, stride in (("conv1", 32000, 3, 2), ("conv2", 15999, 3, 2), ("conv4", 3999, 3, 2), ("conv5", 1999, 2, 2)):
    Ci = Co = 512
    Lout = (Lin - KH) // stride + 1
    x = torch.randn(S, Lin, Ci, device=dev)
    wg = torch.randn(Co, KH * Ci, device=dev) * 0.03
    y, gp = torch.empty(S, Lout, Co, device=dev), torch.empty(S, Lout, Co, device=dev)
    dz = torch.randn(S, Lout, Co, device=dev)
    dwg = torch.empty(Co, KH * Ci, device=dev)
    dx = torch.zeros(S, Lin, Ci, device=dev)
    fl = 2.0 * S * Lout * Co * KH * Ci / 1e12
    r = {"layer": name, "S": S, "rows": S * Lout, "K": KH * Ci}
    r["fwd_gelu_save_TF"] = round(fl / (timeit(lambda: ops.mm_conv_fwd(be, x, wg, y, S, Lin, Lout, Ci, Co, KH, stride, act=MM_ACT_GELU_SAVE_GRAD, pre=gp)) * 1e-3), 1)
    r["fwd_gelu_TF"] = round(fl / (timeit(lambda: ops.mm_conv_fwd(be, x, wg, y, S, Lin, Lout, Ci, Co, KH, stride, act=MM_ACT_GELU)) * 1e-3), 1)
    r["fwd_plain_epilogue_TF"] = round(fl / (timeit(lambda: ops.mm_conv_fwd(be, x, wg, y, S, Lin, Lout, Ci, Co, KH, stride)) * 1e-3), 1)
    r["wgrad_TF"] = round(fl / (timeit(lambda: ops.mm_conv_wgrad(be, x, dz, dwg, S, Lin, Lout, Ci, Co, KH, stride)) * 1e-3), 1)
    r["dgrad_TF"] = round(fl / (timeit(lambda: ops.mm_conv_dgrad(be, dz, wg, dx, S, Lin, Lout, Ci, Co, KH, stride)) * 1e-3), 1)
    # dense proxies: the same M x N x K on plain operands
    M, N, K = S * Lout, Co, KH * Ci
    xd = torch.randn(M, K, device=dev)
    r["dense_fwd_TF"] = round(fl / (timeit(lambda: be.mm(MMOp(xd, K), MMOp(wg, K), y, M, N, K, N)) * 1e-3), 1)
    r["dense_wgrad_TF"] = round(fl / (timeit(lambda: be.mm(MMOp(dz, N, trans=True), MMOp(xd, K, trans=True), dwg, N, K, M, K, split_k=0)) * 1e-3), 1)
    print(json.dumps(r), flush=True)
    del x, y, gp, dz, dx, xd
