"""One short sequence through the HIP path and the float64 oracle: gradient errors per parameter group (real encoder)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import a2f_amd
from test_gpu_models import _faceformer_full, _oracle_ff_run, DEV
rl = lambda a, b: float((a - b).norm() / b.norm())
m, esd, dsd = _faceformer_full()
m.train(); m.audio_encoder.train_regularisers = m.decoder_regularisers = False
for T in (int(v) for v in os.environ.get("TS", "24,30,60,130").split(",")):
    gen = torch.Generator().manual_seed(77)
    audio = (torch.randn(1, T * 16000 // 60 + 8, generator=gen) * 0.1)
    oh = torch.zeros(1, 12); oh[0, 3] = 1
    template = torch.randn(1, 5023, 3, generator=gen) * 5
    gt = template.view(1, 1, 5023, 3) + torch.randn(1, T, 5023, 3, generator=gen) * 0.5
    o64 = _oracle_ff_run(esd, dsd, audio, oh, template, gt, T, torch.float64)
    for scale in (1.0, 1.0 / 9.0):
        m.zero_grad()
        _p, l = m.forward_with_loss_batch(audio.to(DEV), oh.to(DEV), template.to(DEV), gt.to(DEV))
        (l["loss"] * scale).backward()
        g = {k: p.grad.double().cpu() / scale for k, p in m.named_parameters() if p.grad is not None}
        keys = ("vertice_map_r.weight", "transformer_decoder.layers.0.linear1.weight", "transformer_decoder.layers.0.norm3.weight", "obj_vector.weight",
                "audio_encoder.feature_extractor.conv_layers.0.conv.weight", "audio_encoder.encoder.layers.0.attention.v_proj.weight")
        print(f"T={T} loss-scale {scale:.3f}: " + "  ".join(f"{k.split('.')[-2] if 'weight' in k else k}:{rl(g[k], o64['grads'][k]):.1e}" for k in keys))
