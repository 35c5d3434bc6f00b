#!/usr/bin/env python3
"""The Audio2Mesh analysis-stack convolutions as the step runs them (windowed implicit GEMMs on a2f_mm, channels-last, padded channel strides):
forward in bf16x6, weight / data gradient in bf16x3, per layer, beside the time the product's own HBM traffic would take at 6 TB/s.
usage: bench_a2m_conv.py [B]  (A2F_LIB_PATH selects an ablation build of tools/build_variant.sh)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a2f_amd
from a2f_amd import ops, functional as F

dev = torch.device("cuda")
be = ops.backend()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H0 = 64


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


Wd = 16
for l in range(1, 5):
    ci, co, cip, cop = F.A2M_CH[l], F.A2M_CH[l + 1], F.A2M_CHP[l], F.A2M_CHP[l + 1]
    Wo = Wd // 2
    S, R = B * H0, B * H0 * Wo
    x = torch.randn(S, Wd, cip, device=dev)
    wg = torch.randn(cop, 3 * cip, device=dev) * 0.05
    wgT = wg.t().contiguous()
    bias = torch.randn(cop, device=dev)
    z = torch.empty(R, cop, device=dev)
    dz = torch.randn(R, cop, device=dev)
    dwg, dbv = torch.empty(cop, 3 * cip, device=dev), torch.empty(cop, device=dev)
    dx = torch.empty(S * Wd, cip, device=dev)
    r = {"layer": l, "rows": R, "K": 3 * cip, "N": cop}
    mb = (x.numel() + z.numel()) * 4 / 1e6
    r["hbm_floor_us"] = round(mb / 6.0, 1)
    be.mm_precision = 2
    r["fwd_x6_us"] = round(timeit(lambda: ops.mm_conv_fwd(be, x, wg, z, S, Wd, Wo, cip, cop, 3, 2, pad=1, bias=bias)), 1)
    be.mm_precision = 1
    r["fwd_x3_us"] = round(timeit(lambda: ops.mm_conv_fwd(be, x, wg, z, S, Wd, Wo, cip, cop, 3, 2, pad=1, bias=bias)), 1)
    r["wgrad_x3_us"] = round(timeit(lambda: ops.mm_conv_wgrad(be, x, dz, dwg, S, Wd, Wo, cip, cop, 3, 2, pad=1, db=dbv)), 1)
    r["dgrad_x3_us"] = round(timeit(lambda: ops.mm_conv_dgrad(be, dz, wg, dx, S, Wd, Wo, cip, cop, 3, 2, pad=1, WgT=wgT)), 1)
    print(json.dumps(r), flush=True)
    Wd = Wo
    del x, z, dz, dx
