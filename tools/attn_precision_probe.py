#!/usr/bin/env python3
"""configs[4] (FaceFormer, 3000 frames, tolerance 1e-3 on vertex offsets): which attention product form meets the tolerance, and by what margin?

The library's default (bf16x3 GEMMs and attention) sits <= 1e-4 from the fp32 oracle forward and on every gradient (tests/test_gpu_models.py:
test_faceformer_config5_T3000_*), so it is the yardstick here: the same step with plain-bf16 attention products (one MFMA term instead of three, fp32 softmax and
accumulation; a2f_attn_* precision 0) under bf16x3 GEMMs, and with everything plain bf16, is compared with it - vertex offsets, loss, d loss / d audio and every
parameter gradient, on the formula weights and on the well-conditioned weight set, inputs as in the T = 3000 tests.  usage: attn_precision_probe.py [T]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import a2f_amd
from a2f_amd import ops
import test_gpu_models as tg

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
DEV = "cuda"


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def run(m, audio, oh, template, gt, mmp, ap):
    be = ops.backend()
    old = be.mm_precision, be.attn_precision
    be.mm_precision, be.attn_precision = mmp, ap
    try:
        m.zero_grad(set_to_none=True)
        a = audio.to(DEV).requires_grad_(True)
        pred, loss = m.forward_with_loss(a, oh.to(DEV), template.to(DEV), gt.to(DEV))
        loss["loss"].backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        return (pred.detach() - template.to(DEV).view(1, 1, -1, 3)).clone(), float(loss["loss"].detach()), a.grad.detach().clone(), grads
    finally:
        be.mm_precision, be.attn_precision = old


for wname, gains in (("formula weights", None), ("well-conditioned weights", tg.WELL_CONDITIONED_GAINS)):
    m, _esd, _dsd = tg._faceformer_full(gains)
    audio, oh, template, gt = tg._ff_long_inputs(T, 3002)
    ref = run(m, audio, oh, template, gt, 1, None)
    for sname, (mmp, ap) in (("bf16x3 GEMMs + bf16 attention", (1, 0)), ("bf16 everywhere", (0, 0)), ("bf16x3 GEMMs + fp8 attention forward", (1, 2))):
        got = run(m, audio, oh, template, gt, mmp, ap)
        ge = sorted(((rel(got[3][k], ref[3][k]), k) for k in ref[3] if float(ref[3][k].abs().max()) > 0), reverse=True)
        print(f"T = {T}, {wname}, {sname}: vertex offsets {rel(got[0], ref[0]):.2e}, loss {abs(got[1] - ref[1]) / abs(ref[1]):.2e}, d/d audio {rel(got[2], ref[2]):.2e}; "
              f"gradients: worst {ge[0][0]:.2e} ({ge[0][1]}), >1e-3: {sum(e > 1e-3 for e, _ in ge)}, >3.3e-4: {sum(e > 3.3e-4 for e, _ in ge)} of {len(ge)}; "
              f"next worst {[(f'{e:.1e}', k.replace('audio_encoder.encoder.', '')) for e, k in ge[1:6]]}", flush=True)
    del m
    torch.cuda.empty_cache()
