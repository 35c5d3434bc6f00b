"""Gradients of the first feature-extractor layer for one batch computed unsplit, in parts and one sequence at a time (summation-order check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import a2f_amd
from test_gpu_models import _faceformer_full, DEV
S, T = int(os.environ.get("S", 9)), int(os.environ.get("T", 24))
m, _, _ = _faceformer_full()
m.train(); m.audio_encoder.train_regularisers = m.decoder_regularisers = False
gen = torch.Generator().manual_seed(77)
audio = (torch.randn(S, T * 16000 // 60 + 8, generator=gen) * 0.1).to(DEV)
oh = torch.zeros(S, 12); oh[torch.arange(S), torch.arange(S) % 8] = 1
template = torch.randn(S, 5023, 3, generator=gen) * 5
gt = (template.view(S, 1, 5023, 3) + torch.randn(S, T, 5023, 3, generator=gen) * 0.5).to(DEV)
oh, template = oh.to(DEV), template.to(DEV)
def run(cuts):
    m.zero_grad()
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        _p, l = m.forward_with_loss_batch(audio[lo:hi], oh[lo:hi], template[lo:hi], gt[lo:hi])
        (l["loss"] * ((hi - lo) / S)).backward()
    return {k: p.grad.double().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
whole, parts, singles = run([0, S]), run([0, 3, 5, 7, 9] if S == 9 else [0, S // 2, S]), run(list(range(S + 1)))
rl = lambda a, b: float((a - b).norm() / b.norm())
bad = [(rl(whole[k], singles[k]), rl(parts[k], singles[k]), k) for k in whole if float(singles[k].norm()) > 0]
bad.sort(reverse=True)
print("worst whole-vs-singles:")
for e, e2, k in bad[:12]:
    print(f"  {k:75s} {e:.2e} (parts-vs-singles {e2:.2e}) |g| {float(singles[k].norm()):.2e}")
print("decoder-side / head keys:")
for e, e2, k in bad:
    if not k.startswith("audio_encoder."):
        print(f"  {k:75s} {e:.2e} (parts-vs-singles {e2:.2e})")
with torch.no_grad():
    fn = audio.shape[1] * 60 // 16000
    enc_w = m.encode(audio, fn)
    y_w = m.decode_hidden(enc_w, oh)
    enc_s = torch.cat([m.encode(audio[i:i + 1], fn) for i in range(S)])
    y_s = torch.cat([m.decode_hidden(enc_s[i:i + 1], oh[i:i + 1]) for i in range(S)])
    y_ws = torch.cat([m.decode_hidden(enc_w[i:i + 1], oh[i:i + 1]) for i in range(S)])
    print("encoder states batched vs one by one:", [f"{rl(enc_w[i].double(), enc_s[i].double()):.1e}" for i in range(S)])
    print("decoder rows from batched vs from one-by-one encoder states:", [f"{rl(y_w[i].double(), y_s[i].double()):.1e}" for i in range(S)])
    print("decoder rows  batched vs one by one (same encoder states):", [f"{rl(y_w[i].double(), y_ws[i].double()):.1e}" for i in range(S)])
if os.environ.get('ORACLE') != '1':
    sys.exit(0)
# against the CPU oracle (fp32 = the reference's arithmetic, float64 = the truth): the mean over the S sequences of the oracle's per-sequence gradients
from test_gpu_models import _oracle_ff_run
_, esd, dsd = _faceformer_full()
acc = {torch.float32: {}, torch.float64: {}}
for dt in acc:
    for i in range(S):
        o = _oracle_ff_run(esd, dsd, audio[i:i + 1].cpu(), oh[i:i + 1].cpu(), template[i:i + 1].cpu(), gt[i:i + 1].cpu(), T, dt)
        for k, v in o["grads"].items():
            acc[dt][k] = acc[dt].get(k, 0) + v / S
print("key | fp32 oracle vs fp64 | whole vs fp64 | singles vs fp64")
for k in ("vertice_map_r.weight", "transformer_decoder.layers.0.linear1.weight", "transformer_decoder.layers.0.linear2.weight", "transformer_decoder.layers.0.norm3.weight",
          "obj_vector.weight", "audio_feature_map.weight", "audio_encoder.feature_extractor.conv_layers.0.conv.weight", "audio_encoder.encoder.layers.0.attention.v_proj.weight",
          "audio_encoder.encoder.layers.6.feed_forward.intermediate_dense.weight"):
    g64 = acc[torch.float64][k]
    print(f"{k:75s} {rl(acc[torch.float32][k], g64):.2e}  {rl(whole[k], g64):.2e}  {rl(singles[k], g64):.2e}")
