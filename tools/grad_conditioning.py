#!/usr/bin/env python3
"""How well-conditioned are the FaceFormer gradients under a weight set?  Runs the CPU oracle (oracle/wav2vec2.py + decode_incremental + FaceFormerLoss)
at T frames in fp32 and in float64 and prints, per parameter gradient, the relative L2 error of the reference's own fp32 arithmetic against float64.
A gradient the fp32 reference resolves to ~1e-6 is one the 1e-4 parity bar can be held to without a conditioning floor (tests/test_gpu_models.py,
_assert_all_grads); the q / k projections of the late layers under the plain formula weights are not (nearly uniform attention).
usage: grad_conditioning.py T qk_gain ['{"pattern": gain, ...}']"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from oracle import faceformer as o_ff, losses as o_loss, wav2vec2 as o_w2v, weights as W

T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
qk = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
NV = 15069
FF_GAINS = {"vertice_map_r.weight": 4.0, "vertice_map.weight": 0.05, "obj_vector": 2.0, "in_proj_weight": 1.5}      # tests/test_gpu_models.py
gains = {"original0": 40.0}
if qk != 1.0:
    gains.update({"q_proj.weight": qk, "k_proj.weight": qk})
if len(sys.argv) > 3:
    import json
    gains.update(json.loads(sys.argv[3]))
esd = W.formula_state_dict(W.wav2vec2_shapes(), gains=gains)
dsd = W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS)
gen = torch.Generator().manual_seed(600)
audio = torch.randn(1, T * 16000 // 60 + 8, generator=gen) * 0.1
oh = torch.zeros(1, 12); oh[0, 5] = 1
template = torch.randn(1, 5023, 3, generator=gen) * 5
gt = template.view(1, 1, 5023, 3) + torch.randn(1, T, 5023, 3, generator=gen) * 0.5


def run(dtype):
    c = lambda t: t.clone().to(dtype)
    P = {k: (c(v).requires_grad_(True) if v.is_floating_point() and not k.startswith("PPE") else v) for k, v in dsd.items()}
    E = {k: c(v).requires_grad_(True) for k, v in esd.items()}
    a = c(audio).requires_grad_(True)
    enc = o_w2v.wav2vec2_forward(E, o_w2v.normalize_waveform(a), T)
    ref = o_ff.decode_incremental(P, enc, c(oh), c(template))
    ol = o_loss.faceformer_loss(ref, c(gt))
    ol["loss"].backward()
    g = {"audio_encoder." + k: v.grad.double() for k, v in E.items() if v.grad is not None}
    g.update({k: v.grad.double() for k, v in P.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None})
    g["audio"] = a.grad.double()
    return g


t0 = time.time()
g32 = run(torch.float32)
t1 = time.time()
g64 = run(torch.float64)
t2 = time.time()
rows = []
for k, r in g64.items():
    n = float(r.norm())
    if n == 0.0 or k.endswith("k_proj.bias"):           # identically zero in exact arithmetic (softmax shift invariance)
        continue
    rows.append((float((g32[k] - r).norm()) / n, k))
rows.sort(reverse=True)
print(f"T={T} qk_gain={qk}: fp32 {t1 - t0:.0f}s, float64 {t2 - t1:.0f}s; {len(rows)} gradients; worst fp32-vs-float64:")
for e, k in rows[:14]:
    print(f"  {e:.2e}  {k}")
qk_rows = [(e, k) for e, k in rows if "q_proj" in k or "k_proj" in k]
print("  worst q/k:", [(f"{e:.1e}", k.split("layers.")[1]) for e, k in qk_rows[:6]])
print("  d loss / d audio:", [f"{e:.2e}" for e, k in rows if k == "audio"])
print("  count > 1e-6:", sum(1 for e, _ in rows if e > 1e-6), " > 1e-5:", sum(1 for e, _ in rows if e > 1e-5))
