#!/usr/bin/env python3
"""Two runs of one pipelined FaceFormer step from the same state: records, per call of the decoder backward, checksums of its inputs (dy, the save buffer
head) and outputs (dparams, dcross, dstyle) and prints which of them differ between the runs (debugging aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import a2f_amd
import bench
from a2f_amd import ops
from a2f_amd.trainer import TrainStep

S, T = int(os.environ.get("FF_S", 48)), int(os.environ.get("FF_T", 600))
dev = torch.device("cuda")
be = ops.backend()
audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, dev)
log = []
orig_bwd, orig_fwd = be.ff_decoder_bwd, be.ff_decoder_fwd


def bwd(P, dy, S_, T_, period, save, dP, dc, ds, drop=None):
    torch.cuda.current_stream().synchronize()
    rec = {"dy": dy.clone(), "save_rows": save[:S_ * T_ * 896].clone(), "P": P.clone()}
    if drop is not None:
        rec["row"], rec["attn"] = drop[0].clone(), drop[1].clone()
    orig_bwd(P, dy, S_, T_, period, save, dP, dc, ds, drop=drop)
    torch.cuda.current_stream().synchronize()
    off = S_ * T_ * 896 + S_ * 37760 + S_ * T_ * 720 + 64 * (192 * 64 + 192)
    rec.update(dP=dP.clone(), dc=dc.clone(), ds=ds.clone(), nrec=save[off + 128 * 1088 * 0:off + S_ * T_ * 1088].clone(), grec=save[S_ * T_ * 896 + S_ * 37760:S_ * T_ * 896 + S_ * 37760 + S_ * T_ * 720].clone())
    log.append(rec)


be.ff_decoder_bwd = bwd
runs = []
for _run in range(2):
    log.clear()
    torch.manual_seed(1234)
    m = a2f_amd.Faceformer(n_verts=15069, n_onehot=12).to(dev)
    with torch.no_grad():
        m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
    m.train()
    step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
    step.pipeline_min_seqs, step.pipeline_min_frames = 1, 1
    torch.manual_seed(99)
    step(audio, one_hot, gt, templ)
    torch.cuda.synchronize()
    runs.append(list(log))
    del step, m
for h, (a, b) in enumerate(zip(runs[0], runs[1])):
    for k in a:
        x, y = a[k], b[k]
        if k == "nrec":                      # frames of the first block are never written
            x, y = x.view(-1, T, 1088)[:, 128:], y.view(-1, T, 1088)[:, 128:]
        print("half", h, "%-10s equal=%s" % (k, bool(torch.equal(x, y))))
