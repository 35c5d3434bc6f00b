"""Diagnostic: run-to-run differences of the VOCA direct step at the bench size: which of h3 / pred / dH / head dW / db differ, where."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import a2f_amd  # noqa: E402
from a2f_amd.data import feature_windows  # noqa: E402
from a2f_amd.trainer import TrainStep  # noqa: E402

DEV = torch.device("cuda:0")
S, T = 32, 600
g = torch.Generator().manual_seed(21)
feats = torch.randn(S, T, 29, generator=g)
B = S * T
subj = torch.randint(0, 8, (S,), generator=g)
oh = torch.zeros(S, 12); oh[torch.arange(S), subj] = 1.0
oh = oh.repeat_interleave(T, 0).to(DEV)
templ = (torch.randn(8, 5023, 3, generator=g) * 5)[subj].repeat_interleave(T, 0).to(DEV)
gg = torch.Generator(device=DEV).manual_seed(22)
gt = templ + torch.randn(B, 5023, 3, generator=gg, device=DEV) * 0.5
xw = feature_windows(feats).to(DEV)



def fresh(keep):
    torch.manual_seed(5)
    m = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
    st = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-4, use_graph=False)
    torch.cuda.synchronize()
    pred, out3 = m.train_step_direct(xw, oh, templ, gt, 1.0, 10.0, phase=1)
    saved, dH, _ = m._direct_state
    m._direct_state = None
    torch.cuda.synchronize()
    keep.append((m, st))
    addr = {"flat_p": st.opt.flat_p.data_ptr(), "flat_g": st.opt.flat_g.data_ptr(), "W": m.decoder[4].weight.data_ptr(), "b": m.decoder[4].bias.data_ptr(),
            "h3?": saved[6].data_ptr(), "pred": pred.data_ptr(), "dH": dH.data_ptr()}
    return ({"W": m.decoder[4].weight.detach().clone(), "bias": m.decoder[4].bias.detach().clone(), "z2": saved[6].clone(), "pred": pred.reshape(B, -1).clone(), "dH": dH.clone(),
             "dW": m.decoder[4].weight.grad.clone(), "db": m.decoder[4].bias.grad.clone(), "out3": out3.clone()}, addr)


def where(a, b):
    ne = (a != b)
    if a.dim() == 1:
        idx = ne.nonzero().flatten()
        return f"{int(ne.sum())} differ, idx {idx[:6].tolist()}..{idx[-3:].tolist()} max {float((a - b).abs().max()):.3e}" if len(idx) else "equal"
    rows = ne.any(1).nonzero().flatten()
    cols = ne.any(0).nonzero().flatten()
    if len(rows) == 0:
        return "equal"
    return (f"{int(ne.sum())} differ; rows {len(rows)}: {rows[:6].tolist()}..{rows[-3:].tolist()}; cols {len(cols)}: {cols[:6].tolist()}..{cols[-3:].tolist()}; "
            f"max {float((a - b).abs().max()):.3e}")




def fresh0(split):
    torch.manual_seed(5)
    m = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
    st = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-4, use_graph=False)
    torch.cuda.synchronize()
    note = ""
    if split:
        pred, out3 = m.train_step_direct(xw, oh, templ, gt, 1.0, 10.0, phase=1)
        torch.cuda.synchronize()
        dW1, db1 = m.decoder[4].weight.grad.clone(), m.decoder[4].bias.grad.clone()
        m.train_step_direct(xw, oh, templ, gt, 1.0, 10.0, phase=2)
        torch.cuda.synchronize()
        note = f" [phase 2 changed head dW: {not torch.equal(dW1, m.decoder[4].weight.grad)}, db: {not torch.equal(db1, m.decoder[4].bias.grad)}]"
    else:
        pred, out3 = m.train_step_direct(xw, oh, templ, gt, 1.0, 10.0, phase=0)
        torch.cuda.synchronize()
    out = {"pred": pred.reshape(B, -1).clone(), "out3": out3.clone()}
    for k, p in m.named_parameters():
        out[k] = p.grad.detach().clone().reshape(p.shape[0], -1) if p.dim() > 1 else p.grad.detach().clone()
    return out, note


def where(a, b):
    ne = (a != b)
    if a.dim() == 1:
        idx = ne.nonzero().flatten()
        return f"{int(ne.sum())} differ, idx {idx[:4].tolist()}..{idx[-2:].tolist()} max {float((a - b).abs().max()):.3e}" if len(idx) else "equal"
    rows = ne.any(1).nonzero().flatten()
    cols = ne.any(0).nonzero().flatten()
    if len(rows) == 0:
        return "equal"
    return (f"{int(ne.sum())} differ; rows {len(rows)}: {rows[:4].tolist()}..{rows[-2:].tolist()}; cols {len(cols)}: {cols[:4].tolist()}..{cols[-2:].tolist()}; "
            f"max {float((a - b).abs().max()):.3e}")


base, _ = fresh0(False)
bad = 0
N = int(os.environ.get("DIAG_N", "40"))
for i in range(N):
    r, note = fresh0(split=False)
    diff = [k for k in base if not torch.equal(base[k], r[k])]
    if diff:
        bad += 1
        print(f"inst {i}: " + " | ".join(f"{k}: {where(base[k], r[k])}" for k in diff if k.startswith("decoder.4") or k in ("pred", "out3")), flush=True)
print(f"{bad} of {N} fresh instances differ from the first")
