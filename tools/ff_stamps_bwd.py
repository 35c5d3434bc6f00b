#!/usr/bin/env python3
"""Per-phase cycle stamps of ONE step of the decoder backward recurrence (timing build: tools/build_variant.sh with VARIANT_SRC=ff_decoder
-DFF_STAMPS -DFF_STAMP_STEP=<step>): s_memtime after every barrier of the step, read back from sequence 0's gradient slab."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from a2f_amd import ops

S, T, period = int(os.environ.get("FF_S", 32)), int(os.environ.get("FF_T", 600)), 60
dev = torch.device("cuda")
be = ops.backend()
nP = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64 + 6 * 64 + 64 * 64 + 64
g = torch.Generator().manual_seed(0)
P = (torch.randn(nP, generator=g) * 0.1).to(dev)
pe = torch.randn(period, 64, generator=g).to(dev)
cross, style = torch.randn(S, T, 64, generator=g).to(dev), torch.randn(S, 64, generator=g).to(dev)
dy = torch.randn(S, T, 64, generator=g).to(dev)
y = torch.empty(S, T, 64, device=dev)
save = torch.zeros(be.ff_save_floats(S, T), device=dev)
dP, dc, ds = torch.empty(nP, device=dev), torch.empty(S, T, 64, device=dev), torch.empty(S, 64, device=dev)
be.ff_decoder_fwd(P, pe, cross, style, S, T, period, y, save)
be.ff_decoder_bwd(P, dy, S, T, period, save, dP, dc, ds)
torch.cuda.synchronize()
import ctypes
buf = (ctypes.c_longlong * 16)()
be.lib.a2f_debug_ff_bwd_stamps.argtypes = [ctypes.c_void_p]
assert be.lib.a2f_debug_ff_bwd_stamps(buf) == 0
st = [v for v in buf if v != 0]
names = ["", "S1 stage saved rows, LN3'", "S2 W2^T df", "S3 ReLU'", "S4a", "S4 W1^T dhp", "S5 LN2', LN1'", "S6 Wo^T dsa", "S7 attention sweep", "S8 dK/dV row, dq", "S9 Wqkv^T dqkv", "S10 demb, carry"]
print("stamps:", len(st))
for n in range(1, len(st)):
    print("%-28s %6d clk" % (names[n] if n < len(names) else n, st[n] - st[n - 1]))
print("step total:", st[-1] - st[0], "clk")
