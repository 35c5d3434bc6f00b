#!/usr/bin/env python3
"""Throughput of the other two model families on one MI355X (parity-test configurations, not the bench.py line):
  * Audio2Mesh train step (config.yaml batch 128, and 1024)
  * FaceFormer train step: wav2vec2-base encoder (torch ops, fp32) + HIP decoder recurrence + fused head, S sequences x T=600
Prints one JSON object per line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
from a2f_amd.trainer import FlatAdam, TrainStep

dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True          # as the reference harness (Lightning, non-deterministic training) sets it
NV = 15069


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def audio2mesh(B):
    torch.manual_seed(0)
    m = a2f_amd.Audio2Mesh(NV, 12).to(dev).train()
    step = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-4, use_graph=os.environ.get("A2M_GRAPH", "1") == "1")
    x = torch.randn(B, 52, 32, device=dev)
    oh = torch.zeros(B, 12, device=dev); oh[torch.arange(B), torch.arange(B) % 12] = 1
    t = torch.randn(B, 5023, 3, device=dev) * 5
    gt = t + torch.randn(B, 5023, 3, device=dev) * 0.5
    dt = timeit(lambda: step(x, oh, gt, t))
    print(json.dumps({"model": "audio2mesh", "batch": B, "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(B / dt, 1),
                      "mfma_f32_tflops": round(B * 393e6 / dt / 1e12, 2)}), flush=True)


def faceformer(S, T, encoder=True):
    torch.manual_seed(0)
    m = a2f_amd.Faceformer(NV, 12, audio_encoder=None if encoder else torch.nn.Identity()).to(dev)
    amp = os.environ.get("FF_AUTOCAST")                      # "bf16" / "fp16": the reference's `percision: 16-mixed` on the encoder
    if amp:
        m.encoder_autocast = torch.bfloat16 if amp == "bf16" else torch.float16
    with torch.no_grad():                      # non-zero head so the recurrence is exercised
        m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = FlatAdam(params, lr=1e-4, weight_decay=1e-5)
    audio = torch.randn(S, T * 16000 // 60 + 8, device=dev) * 0.1
    oh = torch.zeros(S, 12, device=dev); oh[:, 3] = 1
    t = torch.randn(S, 5023, 3, device=dev) * 5
    gt = t.view(S, 1, 5023, 3) + torch.randn(S, T, 5023, 3, device=dev) * 0.5
    enc_fixed = torch.randn(S, T, 768, device=dev)

    def step():
        opt.zero_grad()
        if encoder:
            _, loss = m.forward_with_loss_batch(audio, oh, t, gt)
        else:
            y = m.decode_hidden(enc_fixed, oh)
            from a2f_amd.functional import HeadLossFusedFn
            tot = None
            for s in range(S):
                _, o3 = HeadLossFusedFn.apply(y[s], m.vertice_map_r.weight, m.vertice_map_r.bias, t[s].reshape(1, -1), gt[s].reshape(T, -1), T - (T % 2), 1.0, 10.0)
                tot = o3 if tot is None else tot + o3
            loss = {"loss": tot[0] / S}
        loss["loss"].backward()
        opt.step()
    dt = timeit(step, warm=2, iters=5)
    print(json.dumps({"model": "faceformer" + ("" if encoder else "-decoder+head-only") + (f" (encoder autocast {amp})" if amp else ""), "sequences": S, "T": T, "ms_per_step": round(dt * 1e3, 2),
                      "frames_per_s": round(S * T / dt, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "audio2mesh":      # e.g. under rocprofv3: one configuration only
        audio2mesh(int(sys.argv[2]))
        sys.exit(0)
    if len(sys.argv) >= 3 and sys.argv[1] == "faceformer":
        faceformer(int(sys.argv[2]), 600, encoder=len(sys.argv) < 4)
        sys.exit(0)
    audio2mesh(128)
    audio2mesh(1024)
    faceformer(8, 600, encoder=False)
    faceformer(64, 600, encoder=False)
    faceformer(4, 600, encoder=True)
    faceformer(16, 600, encoder=True)
