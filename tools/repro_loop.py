#!/usr/bin/env python3
"""Diagnostic: the body of tests/test_gpu_models.py::test_faceformer_full_size_pipelined_step_is_bitwise_reproducible N times in one process - two runs of two
train-mode steps from the same state and seeds must leave bitwise equal parameters; a rare race needs many repetitions, and the report names the parameters
that differ.  usage: repro_loop.py [N] [seqs] [frames]      env: REPRO_PIPELINE=0 (one stream), A2F_WGRAD_STREAM=0 (no weight-gradient lane), ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
import bench
from a2f_amd.trainer import TrainStep

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
pipeline = os.environ.get("REPRO_PIPELINE", "1") != "0"
dev = torch.device("cuda")
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
fails = 0
for rep in range(N):
    finals = []
    for _run in range(2):
        torch.manual_seed(1234)
        m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
        with torch.no_grad():
            m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
        m.train()
        if os.environ.get("REPRO_NO_REGULARISERS") == "1":          # the deterministic function in train mode: no dropout-variant kernels (the only scratch users of the step)
            m.audio_encoder.train_regularisers = m.decoder_regularisers = False
        step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
        step.pipeline = pipeline
        step.pipeline_min_seqs, step.pipeline_min_frames = 1, 1
        torch.manual_seed(99)
        for _ in range(2):
            out = step(audio, one_hot, gt, templ)
        torch.cuda.synchronize()
        spans = {name: step.opt._offsets[id(p)] for name, p in m.named_parameters() if id(p) in step.opt._offsets}
        finals.append((step.opt.flat_p.clone(), float(out["loss"].detach())))
        step.close()
        del step, m
    if finals[0][1] != finals[1][1] or not torch.equal(finals[0][0], finals[1][0]):
        fails += 1
        d = (finals[0][0] - finals[1][0]).abs()
        bad = [(name, float(d[o:o + n].max())) for name, (o, n) in spans.items() if float(d[o:o + n].max()) > 0]
        print(f"rep {rep}: loss {finals[0][1]!r} / {finals[1][1]!r}; {len(bad)} of {len(spans)} parameters differ; last: "
              + ", ".join(f"{k.replace('audio_encoder.encoder.', '')} {v:.1e}" for k, v in bad[-8:]), flush=True)
print(f"fails {fails} of {N}  (S = {S}, T = {T}, pipeline = {pipeline}, lane = {os.environ.get('A2F_WGRAD_STREAM', '1')})", flush=True)
