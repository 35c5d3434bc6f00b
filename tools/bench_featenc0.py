#!/usr/bin/env python3
"""Layer 0 of the wav2vec2 feature encoder (csrc/featenc.hip: convolution recomputed in every pass) at the half-batch shape of configs[2]: 32 sequences x 160 008
samples -> (32, 32 000, 512).  Forward (statistics + apply, image output) and backward (reduce + apply with dy) in ms, beside the output's bytes at 6 TB/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import a2f_amd  # noqa: F401
from a2f_amd import ops

be = ops.backend()
dev = torch.device("cuda")
S, N, C = int(os.environ.get("FE_S", "32")), 160008, 512
L = (N - 10) // 5 + 1
g = torch.Generator().manual_seed(0)
wave = torch.randn(S, N, generator=g).to(dev)
w, gam, bet = (torch.randn(C, 10, generator=g) * 0.3).to(dev), torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
out, stats = torch.empty(S, L, C, device=dev), torch.empty(S, 2, C, device=dev)
dout, dy = torch.randn(S, L, C, device=dev), torch.empty(S, L, C, device=dev)
dw, dg, db = torch.empty(C, 10, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)


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


fwd = timeit(lambda: be.featenc0_fwd(wave, S, N, w, gam, bet, C, 1e-5, out, stats, out_img=True))
bwd = timeit(lambda: be.featenc0_bwd(wave, S, N, w, gam, bet, C, stats, dout, dw, dg, db, dy))
gb = S * L * C * 4 / 1e9
print(json.dumps({"S": S, "L": L, "C": C, "fwd_ms": round(fwd, 4), "bwd_ms": round(bwd, 4), "output_GB": round(gb, 3), "one_pass_at_6TBs_ms": round(gb / 6.0, 4)}))
