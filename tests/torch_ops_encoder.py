"""TEST INFRASTRUCTURE: a plain torch-ops evaluation of ``a2f_amd.model.wav2vec.Wav2Vec2Model`` (HF wav2vec2-base as the reference
wraps it, /root/reference/src/model/wav2vec.py:87-189) on the module's own parameters.  The product module only has the
hand-written HIP path; this restatement is what the CPU suite compares that path (through the C-ABI emulator) and the
HF-generated fixtures (G6, G13) against.  Never imported by the package.
"""
from __future__ import annotations

import types

import torch
import torch.nn.functional as F

from a2f_amd.dropout import unpack_keep_bits
from a2f_amd.model.wav2vec import CONV_STRIDE


def _keep(x, entry):
    """entry = (bits, scale): x * keep * scale."""
    bits, scale = entry[0], entry[1]
    return x * (unpack_keep_bits(bits, x.shape[-1]).view(x.shape) * scale).to(x.dtype)


def feature_extractor(m, wave):
    z = wave[:, None, :]
    for i, layer in enumerate(m.feature_extractor.conv_layers):
        z = F.conv1d(z, layer.conv.weight, None, stride=CONV_STRIDE[i])
        if i == 0:
            z = F.group_norm(z, z.shape[1], layer.layer_norm.weight, layer.layer_norm.bias, 1e-5)
        z = F.gelu(z)
    return z


def encoder_layer(layer, h, masks=None):
    B, T, d = h.shape
    H = layer.heads
    hd = d // H
    a = layer.attention
    q = a.q_proj(h).view(B, T, H, hd).transpose(1, 2)
    k = a.k_proj(h).view(B, T, H, hd).transpose(1, 2)
    v = a.v_proj(h).view(B, T, H, hd).transpose(1, 2)
    mk = masks or {}
    drop = lambda x, key: _keep(x, mk[key]) if key in mk else x
    if "attn" in mk:                                                    # HF eager attention: softmax -> dropout -> @ v
        pr = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
        pr = pr * (unpack_keep_bits(mk["attn"][0], T).view(B, H, T, T) * mk["attn"][2]).to(pr.dtype)
        ctx = torch.matmul(pr, v)
    else:
        ctx = F.scaled_dot_product_attention(q, k, v)
    h = layer.layer_norm(h + drop(a.out_proj(ctx.transpose(1, 2).reshape(B, T, d)), "hid1"))
    f = layer.feed_forward
    return layer.final_layer_norm(h + drop(f.output_dense(drop(F.gelu(f.intermediate_dense(h)), "act")), "hid2"))


def encoder(m, h, masks=None):
    enc = m.encoder
    pc = enc.pos_conv_embed
    z = pc.conv(h.transpose(1, 2))
    z = z[:, :, :-1] if pc.k % 2 == 0 else z
    h = enc.layer_norm(h + F.gelu(z).transpose(1, 2))
    if masks is not None and masks.enc is not None:
        h = _keep(h, masks.enc)
    for l, layer in enumerate(enc.layers):
        if masks is not None and masks.layer_skip[l]:
            continue                                                     # LayerDrop
        h = encoder_layer(layer, h, masks.layers[l] if masks is not None else None)
    return h


def forward(m, input_values, dataset="vocaset", frame_num=None):
    """The wrapper's forward (vocaset branch: 49 Hz features interpolated to ``frame_num`` frames, align_corners=True)."""
    if dataset is None:
        frame_num = None
    z = feature_extractor(m, input_values).transpose(1, 2)
    if frame_num is not None:
        z = F.interpolate(z.transpose(1, 2), size=frame_num, align_corners=True, mode="linear").transpose(1, 2)
    fp = m.feature_projection
    h = fp.projection(fp.layer_norm(z))
    masks = m.last_masks = m._step_masks(h.shape[0], h.shape[1], h.device)
    if masks is not None and masks.featproj is not None:
        h = _keep(h, masks.featproj)
    h = m._spec_augment(h, masks)
    return types.SimpleNamespace(last_hidden_state=encoder(m, h, masks))


class TorchOpsEncoder(torch.nn.Module):
    """Wraps a Wav2Vec2Model so that calling it evaluates the torch-ops restatement (for Faceformer(audio_encoder=...) in CPU tests)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, input_values, dataset="vocaset", attention_mask=None, frame_num=None, **_):
        return forward(self.model, input_values, dataset, frame_num)
