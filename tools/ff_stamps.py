#!/usr/bin/env python3
"""Per-phase cycle stamps of ONE step of the decoder forward (timing build: tools/build_variant.sh with VARIANT_SRC=ff_decoder
-DFF_STAMPS -DFF_STAMP_STEP=<step>, loaded through A2F_LIB_PATH): s_memtime after every barrier of the step, read back from the
A / c block of sequence 0's gradient slab."""
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
y = torch.empty(S, T, 64, device=dev)
save = torch.zeros(be.ff_save_floats(S, T), device=dev)
drop = None
if os.environ.get("FF_DROP") == "1":          # the train-mode variant (keep bits for every dropout site)
    row = torch.empty(S * T, 10, dtype=torch.int32, device=dev); attn = torch.empty(S * 4 * T, be.attn_keep_words(T), dtype=torch.int32, device=dev)
    be.keep_bits(1, 1000, S * T, 320, 0.1, row); be.keep_bits(1, 1001, S * 4 * T, T, 0.1, attn)
    drop = (row, attn, 1 / 0.9)
for _ in range(2):
    be.ff_decoder_fwd(P, pe, cross, style, S, T, period, y, save, drop=drop)
torch.cuda.synchronize()
off = S * T * 896 + 33600
st = save[off:off + 64].view(torch.int64).cpu().tolist()
st = [v for v in st if v != 0]
print("stamps:", len(st))
names = ["top", "x ready", "q|k|v", "sweep", "merge", "out-proj+LN1+LN2", "linear1", "linear2 partials", "LN3+y", "(next top)"]
for n in range(1, len(st)):
    print("%-20s %6d clk" % (names[n] if n < len(names) else n, st[n] - st[n - 1]))
print("step total (stamp 0 .. last):", st[-1] - st[0], "clk  (s_memtime ticks: 100 MHz constant clock on gfx9 -> x10 ns)")
