#!/usr/bin/env python3
"""GPU probe behind tests/test_gpu_models.py::test_faceformer_config2_full_size_*: for a list of encoder weight-gain sets, the HIP path's error on every gradient
of the full-size step (T = 600) against the fp32 and float64 oracles.  usage: grad_parity_probe.py '<json gains>' ['<json gains>' ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_models as tg
from a2f_amd import encoder_ops
from golden_util import rel_l2

encoder_ops.IMG_MIN_ROWS = int(os.environ.get("PROBE_IMG_MIN_ROWS", "0"))
T = 600
for spec in sys.argv[1:]:
    gains = {"original0": 40.0}
    gains.update(json.loads(spec))
    seed, subj = int(gains.pop("_seed", 601)), int(gains.pop("_subj", 2))
    m, esd, dsd = tg._faceformer_full(gains)
    gen = torch.Generator().manual_seed(seed)
    audio = torch.randn(1, T * 16000 // 60 + 8, generator=gen) * 0.1
    oh = torch.zeros(1, 12); oh[0, subj] = 1
    template = torch.randn(1, 5023, 3, generator=gen) * 5
    gt = template.view(1, 1, 5023, 3) + torch.randn(1, T, 5023, 3, generator=gen) * 0.5
    o32 = tg._oracle_ff_run(esd, dsd, audio, oh, template, gt, T, torch.float32)
    o64 = tg._oracle_ff_run(esd, dsd, audio, oh, template, gt, T, torch.float64)
    a = audio.to("cuda").requires_grad_(True)
    pred, loss = m.forward_with_loss(a, oh.cuda(), template.cuda(), gt.cuda())
    loss["loss"].backward()
    params = dict(m.named_parameters())
    rows = []
    for k, r in o64["grads"].items():
        if k in ("audio", "enc") or k not in params or k.endswith("k_proj.bias") or float(r.abs().max()) == 0.0:
            continue
        rows.append((rel_l2(params[k].grad, r), rel_l2(o32["grads"][k], r), k))
    rows.sort(reverse=True)
    bad = [x for x in rows if x[0] >= 1e-4]
    print(json.dumps({"gains": spec, "pred_err": rel_l2(pred.detach().cpu() - template.view(1, 1, 5023, 3), o32["pred"] - template.view(1, 1, 5023, 3)),
                      "audio_grad_err": rel_l2(a.grad, o64["grads"]["audio"]), "n_over_1e-4": len(bad),
                      "worst": [(f"{h:.1e}", f"{f:.1e}", k[-48:]) for h, f, k in rows[:4]], "median_hip_err": f"{rows[len(rows) // 2][0]:.1e}",
                      "decoder": [(f"{h:.1e}", k[-40:]) for h, f, k in rows if not k.startswith("audio_encoder")][:4]}), flush=True)
    del m
    torch.cuda.empty_cache()
