#!/usr/bin/env python3
"""Diagnostic: out-of-bounds WRITES of the kernels that run on the pipelined step's side stream (decoder recurrence, head, decoder-side products).  Every
torch.empty / empty_like / zeros made while a side stream is current gets 4 KB of sentinel bytes behind it, held until the step is over and checked.
usage: guard_probe.py [seqs] [frames] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

GUARD = 4096
_empty = torch.empty
held = []
ALL = os.environ.get("GUARD_ALL") == "1"        # also the main stream's allocations (memory: every buffer of the step stays alive)


def guarded_empty(*shape, **kw):
    dev = kw.get("device")
    if dev is None or torch.device(dev).type != "cuda":
        return _empty(*shape, **kw)
    if not ALL and torch.cuda.current_stream().cuda_stream == 0:
        return _empty(*shape, **kw)
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    dtype = kw.get("dtype") or torch.get_default_dtype()
    n = 1
    for s in shape:
        n *= int(s)
    item = torch.empty((), dtype=dtype).element_size()
    nbytes = (n * item + 15) // 16 * 16
    raw = _empty(nbytes + GUARD, dtype=torch.uint8, device=dev)
    raw[nbytes:].fill_(0xA5)
    held.append((raw, nbytes, tuple(shape), torch.cuda.current_stream().cuda_stream))
    return raw[: n * item].view(dtype).view(*shape) if n else _empty(*shape, **kw)


torch.empty = guarded_empty

import a2f_amd
import bench
from a2f_amd.trainer import TrainStep

S = int(sys.argv[1]) if len(sys.argv) > 1 else 48
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda")
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
torch.manual_seed(1234)
m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
with torch.no_grad():
    m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
m.train()
if os.environ.get("REPRO_NO_REGULARISERS") == "1":
    m.audio_encoder.train_regularisers = m.decoder_regularisers = False
step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
step.pipeline_min_seqs, step.pipeline_min_frames = 0, 1
torch.manual_seed(99)
for it in range(steps):
    held.clear()
    out = step(audio, one_hot, gt, templ)
    torch.cuda.synchronize()
    bad = []
    for raw, nbytes, shape, st in held:
        tail = raw[nbytes:]
        if not bool((tail == 0xA5).all()):
            first = int((tail != 0xA5).nonzero()[0])
            bad.append((shape, st, first, int((tail != 0xA5).sum())))
    if bad or it < 2 or it == steps - 1:
      print(f"step {it}: loss {float(out['loss'].detach()):.6f}; guarded allocations {len(held)}, {sum(r[0].numel() for r in held) / 2**30:.1f} GiB; overwritten guards: {len(bad)}", bad[:20], flush=True)
