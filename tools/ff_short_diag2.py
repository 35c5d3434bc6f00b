"""Every sequence of the S = 9, T = 24 batch on its own: HIP gradient vs float64 oracle gradient (which sequence, if any, is off?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import a2f_amd
from test_gpu_models import _faceformer_full, _oracle_ff_run, DEV
rl = lambda a, b: float((a - b).norm() / b.norm())
S, T = 9, 24
m, esd, dsd = _faceformer_full()
m.train(); m.audio_encoder.train_regularisers = m.decoder_regularisers = False
gen = torch.Generator().manual_seed(77)
audio = (torch.randn(S, T * 16000 // 60 + 8, generator=gen) * 0.1)
oh = torch.zeros(S, 12); oh[torch.arange(S), torch.arange(S) % 8] = 1
template = torch.randn(S, 5023, 3, generator=gen) * 5
gt = (template.view(S, 1, 5023, 3) + torch.randn(S, T, 5023, 3, generator=gen) * 0.5)
key = "transformer_decoder.layers.0.linear1.weight"
tot_h, tot_o = 0, 0
for s in range(S):
    o64 = _oracle_ff_run(esd, dsd, audio[s:s + 1], oh[s:s + 1], template[s:s + 1], gt[s:s + 1], T, torch.float64)
    m.zero_grad()
    _p, l = m.forward_with_loss_batch(audio[s:s + 1].to(DEV), oh[s:s + 1].to(DEV), template[s:s + 1].to(DEV), gt[s:s + 1].to(DEV))
    l["loss"].backward()
    g = m.transformer_decoder.layers[0].linear1.weight.grad.double().cpu()
    tot_h, tot_o = tot_h + g, tot_o + o64["grads"][key]
    print(f"sequence {s}: HIP vs float64 {rl(g, o64['grads'][key]):.2e}   |g| {float(g.norm()):.3e}  loss {float(l['loss']):.5f} vs {o64['loss']:.5f}")
print("sum over sequences: HIP vs float64", f"{rl(tot_h, tot_o):.2e}")
# now accumulate in .grad across the passes, as the parts do
m.zero_grad()
for s in range(S):
    _p, l = m.forward_with_loss_batch(audio[s:s + 1].to(DEV), oh[s:s + 1].to(DEV), template[s:s + 1].to(DEV), gt[s:s + 1].to(DEV))
    l["loss"].backward()
print("accumulated in .grad: vs float64 sum", f"{rl(m.transformer_decoder.layers[0].linear1.weight.grad.double().cpu(), tot_o):.2e}", " vs the HIP sum of separate passes", f"{rl(m.transformer_decoder.layers[0].linear1.weight.grad.double().cpu(), tot_h):.2e}")
