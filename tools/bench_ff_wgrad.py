#!/usr/bin/env python3
"""Timing of the five weight-gradient products the record-form decoder backward issues (strided operands inside the gradient records /
save rows), for a few split-K factors and tile configurations (A2F_MM_CFG)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from a2f_amd import ops
from a2f_amd.ops import MMOp

dev = torch.device("cuda")
be = ops.backend()
R = int(os.environ.get("ROWS", 38400))
grec = torch.randn(R, 720, device=dev)
save = torch.randn(R, 896, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, N, a_off, b_off, b_t in ((192, 64, 0, 0, save), (64, 64, 192, 128, save), (128, 64, 256, 512, grec), (64, 128, 384, 320, save)):
    for split in (8, 16, 32, 64):
        dW, db = torch.empty(M, N, device=dev), torch.empty(M, device=dev)
        A = MMOp(grec, 720, trans=True, ptr_off=a_off)
        B = MMOp(b_t, b_t.shape[1], trans=True, ptr_off=b_off)
        us = timed(lambda: be.mm(A, B, dW, M, N, R, N, split_k=split, a_rowsum=db))
        print(json.dumps({"M": M, "N": N, "K": R, "split_k": split, "cfg": os.environ.get("A2F_MM_CFG", "auto"), "us": round(us, 1)}), flush=True)
