"""Prediction and per-parameter gradient error of Audio2Mesh against the G2 golden fixture (train mode), e.g. to compare
A2F_A2M_MM_PRECISION=2 (bf16x6, the default) with 1 (bf16x3 split products)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, a2f_amd
from golden_util import load, packed_error
from oracle import weights as W
NV=5023*3
g = load("g2_audio2mesh")
for mode in (os.environ.get("A2M_MODE", "train"),):
    m = a2f_amd.Audio2Mesh(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.audio2mesh_shapes(NV, 12), gains={"output_net.4": 4.0}))
    m.to("cuda").train(mode == "train")
    B = 4
    x = W.formula_input("g2.x", (B, 52, 32)).cuda()
    one_hot = torch.from_numpy(g["one_hot"]).cuda()
    template = W.formula_input("g2.template", (B, 5023, 3), 5.0)
    gt = (template + W.formula_input("g2.gt", (B, 5023, 3), 0.5)).cuda()
    pred = m(x, one_hot, template.cuda())
    print("pred err", packed_error(g, f"{mode}::pred", pred))
    loss = a2f_amd.VocaLoss()(pred, gt)
    loss["loss"].backward()
    cancelled = {f"analysis_net.{3 * l}.bias" for l in range(5)} | {f"articulation_net.{i}.bias" for i in (0, 3, 6)}     # exact gradient 0 under batch-stat BN
    for k, p in m.named_parameters():
        if k not in cancelled:
            print("%-40s %.2e" % (k, packed_error(g, f"{mode}::grad::{k}", p.grad)))
