"""wav2vec2-base encoder as used by FaceFormer, CPU restatement (test infrastructure).

The arithmetic belongs to HF ``transformers`` (third party, not under /root/reference, version
un-pinned by the reference; build container has 5.15.0).  Call sites restated here:

* waveform normalisation   <- Wav2Vec2FeatureExtractor(do_normalize=True) called at faceformer.py:142-144
* ``linear_interpolation`` <- /root/reference/src/model/wav2vec.py:76-84 (align_corners=True)
* ``wav2vec2_forward``     <- /root/reference/src/model/wav2vec.py:91-189, "vocaset" branch, eval mode
  (SpecAugment at :149-173 is train-only), which calls HF's Wav2Vec2FeatureEncoder (7 conv layers,
  GroupNorm on layer 0, GELU), Wav2Vec2FeatureProjection (LayerNorm -> Linear), Wav2Vec2Encoder
  (weight-normed grouped positional conv + GELU, LayerNorm, 12 post-norm layers).

Published algorithm: Baevski et al., "wav2vec 2.0", NeurIPS 2020; config = transformers'
default ``Wav2Vec2Config()`` (== facebook/wav2vec2-base-960h architecture).
Parity is pinned against the installed HF module with formula weights (tests/golden/gen_golden.py, G6).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
N_LAYERS = 12
N_HEADS = 12
POS_K = 128
POS_GROUPS = 16
EPS = 1e-5


def normalize_waveform(audio: torch.Tensor) -> torch.Tensor:
    """Zero-mean / unit-variance per utterance: (x - mean) / sqrt(var + 1e-7) (population variance)."""
    x = audio if audio.dtype == torch.float64 else audio.to(torch.float32)      # float64 stays float64 (the tests' high-precision reference runs)
    m = x.mean(dim=-1, keepdim=True)
    v = ((x - m) ** 2).mean(dim=-1, keepdim=True)
    return (x - m) / torch.sqrt(v + 1e-7)


def linear_interpolation(features: torch.Tensor, output_len: int) -> torch.Tensor:
    """(1,L,C) -> (1,output_len,C), F.interpolate(mode='linear', align_corners=True).  wav2vec.py:76-84."""
    y = F.interpolate(features.transpose(1, 2), size=output_len, align_corners=True, mode="linear")
    return y.transpose(1, 2)


def feature_encoder(P: Dict[str, torch.Tensor], wave: torch.Tensor) -> torch.Tensor:
    """(1,N) -> (1,512,L)."""
    z = wave[:, None, :]
    for i, (k, s) in enumerate(zip(CONV_KERNEL, CONV_STRIDE)):
        z = F.conv1d(z, P[f"feature_extractor.conv_layers.{i}.conv.weight"], None, stride=s)
        if i == 0:
            c = z.shape[1]
            z = F.group_norm(z, c, P["feature_extractor.conv_layers.0.layer_norm.weight"],
                             P["feature_extractor.conv_layers.0.layer_norm.bias"], EPS)
        z = F.gelu(z)
    return z


def positional_conv(P, h: torch.Tensor) -> torch.Tensor:
    """Weight-normalised (dim=2) grouped Conv1d(768,768,k=128,pad=64,groups=16), drop last frame (even kernel), GELU."""
    g = P["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = P["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    w = v * (g / v.norm(p=2, dim=(0, 1), keepdim=True))
    z = F.conv1d(h.transpose(1, 2), w, P["encoder.pos_conv_embed.conv.bias"], padding=POS_K // 2, groups=POS_GROUPS)
    z = z[:, :, :-1] if POS_K % 2 == 0 else z
    return F.gelu(z).transpose(1, 2)


def encoder_layer(P, pre: str, h: torch.Tensor) -> torch.Tensor:
    """Post-norm Transformer layer: attn -> +res -> LN -> FFN(GELU) -> +res -> LN."""
    T, d = h.shape[1], h.shape[2]
    hd = d // N_HEADS
    x = h[0]
    q = F.linear(x, P[pre + "attention.q_proj.weight"], P[pre + "attention.q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(x, P[pre + "attention.k_proj.weight"], P[pre + "attention.k_proj.bias"])
    v = F.linear(x, P[pre + "attention.v_proj.weight"], P[pre + "attention.v_proj.bias"])
    q, k, v = (t.view(T, N_HEADS, hd).transpose(0, 1) for t in (q, k, v))
    a = torch.softmax(torch.matmul(q, k.transpose(1, 2)), dim=-1)
    ctx = torch.matmul(a, v).transpose(0, 1).reshape(T, d)
    x = x + F.linear(ctx, P[pre + "attention.out_proj.weight"], P[pre + "attention.out_proj.bias"])
    x = F.layer_norm(x, (d,), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], EPS)
    f = F.linear(F.gelu(F.linear(x, P[pre + "feed_forward.intermediate_dense.weight"], P[pre + "feed_forward.intermediate_dense.bias"])),
                 P[pre + "feed_forward.output_dense.weight"], P[pre + "feed_forward.output_dense.bias"])
    x = F.layer_norm(x + f, (d,), P[pre + "final_layer_norm.weight"], P[pre + "final_layer_norm.bias"], EPS)
    return x[None]


def wav2vec2_forward(P: Dict[str, torch.Tensor], wave: torch.Tensor, frame_num: int, n_layers: int = N_LAYERS, checkpoint: bool = False) -> torch.Tensor:
    """Normalised waveform (1,N) -> last_hidden_state (1,frame_num,768).  wav2vec.py:116-189 (eval).  ``checkpoint``: recompute each
    Transformer layer in the backward instead of keeping its (12, T, T) attention matrices (same arithmetic; bounds the host memory of
    the T = 3000 gradient tests)."""
    z = feature_encoder(P, wave).transpose(1, 2)                                   # (1,L,512)  :116-117
    z = linear_interpolation(z, frame_num)                                         # :125-128
    c = z.shape[-1]
    z = F.layer_norm(z, (c,), P["feature_projection.layer_norm.weight"], P["feature_projection.layer_norm.bias"], EPS)
    h = F.linear(z, P["feature_projection.projection.weight"], P["feature_projection.projection.bias"])   # :147
    h = h + positional_conv(P, h)
    d = h.shape[-1]
    h = F.layer_norm(h, (d,), P["encoder.layer_norm.weight"], P["encoder.layer_norm.bias"], EPS)
    for l in range(n_layers):
        if checkpoint and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint as _ckpt
            h = _ckpt(lambda hh, pre=f"encoder.layers.{l}.": encoder_layer(P, pre, hh), h, use_reentrant=False)
        else:
            h = encoder_layer(P, f"encoder.layers.{l}.", h)
    return h
