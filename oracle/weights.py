"""Formula-generated parameters (test infrastructure, see oracle/__init__.py).

Weights are a pure arithmetic function of (parameter name, shape): no RNG, no
dependence on torch version, so the reference (in the build container), the
oracle and the HIP product can all be loaded with bit-identical values through
``load_state_dict`` - which also checks state-dict key compatibility with the
reference modules.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _fan_in(shape: Tuple[int, ...]) -> int:
    if len(shape) <= 1:
        return max(int(shape[0]) if shape else 1, 1)
    n = 1
    for s in shape[1:]:
        n *= int(s)
    return max(n, 1)


def formula_tensor(name: str, shape: Tuple[int, ...], gain: float = 1.0) -> torch.Tensor:
    """Deterministic pseudo-random-looking values in [-a, a], a = gain*sqrt(3/fan_in).

    value[i] = a * sin(phi(i)), phi a fast-growing quadratic phase so neighbouring
    elements decorrelate; float64 maths rounded once to float32.
    """
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    seed = zlib.crc32(name.encode("utf-8")) & 0xFFFF
    i = np.arange(n, dtype=np.float64)
    phase = 0.6180339887498949 * (i + 1.0) * (1.0 + (seed % 97)) + 1e-3 * np.mod(i * i, 8191.0) + seed * 0.001
    amp = gain * np.sqrt(3.0 / _fan_in(shape))
    lname = name.lower()
    if lname.endswith("bias") or ".bias" in lname:
        amp = 0.05 * gain
    v = amp * np.sin(phase)
    if ("norm" in lname or "bn" in lname) and lname.endswith("weight") and len(shape) == 1:
        v = 1.0 + 0.1 * np.sin(phase)          # affine scales near one
    if lname.endswith("running_var"):
        v = 1.0 + 0.25 * np.sin(phase) ** 2
    if lname.endswith("running_mean"):
        v = 0.1 * np.sin(phase)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def formula_state_dict(shapes: Dict[str, Tuple[int, ...]], gains: Dict[str, float] | None = None) -> Dict[str, torch.Tensor]:
    gains = gains or {}
    out = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
            continue
        g = 1.0
        for pat, gv in gains.items():
            if pat in k:
                g = gv
        out[k] = formula_tensor(k, shp, g)
    return out


def shapes_of(module_or_sd) -> Dict[str, Tuple[int, ...]]:
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: tuple(v.shape) for k, v in sd.items()}


def formula_input(name: str, shape: Iterable[int], scale: float = 1.0) -> torch.Tensor:
    """Deterministic input tensor (same generator, unit-ish amplitude)."""
    shape = tuple(int(s) for s in shape)
    t = formula_tensor("input::" + name, (1,) + shape, 1.0)  # fan_in = numel -> rescale below
    n = int(np.prod(shape))
    return (t.reshape(shape) * float(np.sqrt(n / 3.0)) * scale).contiguous()


# ---- reference-compatible shape tables (so oracle tests need no reference import) ----

def voca_shapes(n_verts: int = 15069) -> Dict[str, Tuple[int, ...]]:
    """Keys/shapes of reference Voca.state_dict() (voca.py:19-36)."""
    s = {}
    for idx, (ci, co) in zip((0, 2, 4, 6), ((37, 32), (32, 32), (32, 64), (64, 64))):
        s[f"time_conv.{idx}.weight"] = (co, ci, 3, 1)
        s[f"time_conv.{idx}.bias"] = (co,)
    for idx, (ci, co) in zip((0, 1, 3, 4), ((72, 72), (72, 128), (128, 50), (50, n_verts))):
        s[f"decoder.{idx}.weight"] = (co, ci)
        s[f"decoder.{idx}.bias"] = (co,)
    return s


def audio2mesh_shapes(n_verts: int = 15069, n_onehot: int = 12) -> Dict[str, Tuple[int, ...]]:
    """Keys/shapes of reference Audio2Mesh.state_dict() (audio2face.py:13-55)."""
    s = {}
    chans = (1, 72, 108, 162, 243, 256)
    for l in range(5):
        s[f"analysis_net.{3 * l}.weight"] = (chans[l + 1], chans[l], 1, 3)
        s[f"analysis_net.{3 * l}.bias"] = (chans[l + 1],)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"analysis_net.{3 * l + 1}.{nm}"] = (chans[l + 1],)
        s[f"analysis_net.{3 * l + 1}.num_batches_tracked"] = ()
    # articulation_net indices: conv 0,3,6 with BN 1,4,7; then BN 9 conv 10; BN 12 conv 13
    for ci, bi in ((0, 1), (3, 4), (6, 7), (10, 9), (13, 12)):
        k = 4 if ci == 13 else 3
        s[f"articulation_net.{ci}.weight"] = (256, 256, k, 1)
        s[f"articulation_net.{ci}.bias"] = (256,)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"articulation_net.{bi}.{nm}"] = (256,)
        s[f"articulation_net.{bi}.num_batches_tracked"] = ()
    for idx, (ci, co) in zip((0, 1, 3, 4), ((256 + n_onehot, 72), (72, 128), (128, 50), (50, n_verts))):
        s[f"output_net.{idx}.weight"] = (co, ci)
        s[f"output_net.{idx}.bias"] = (co,)
    return s


def faceformer_decoder_shapes(n_verts: int = 15069, n_onehot: int = 12, d: int = 64) -> Dict[str, Tuple[int, ...]]:
    """Decoder-side keys/shapes of reference Faceformer.state_dict() (faceformer.py:112-131).

    The ``PPE.pe`` buffer and ``audio_encoder.*`` are excluded (closed form / separate table).
    """
    s = {
        "audio_feature_map.weight": (d, 768), "audio_feature_map.bias": (d,),
        "vertice_map.weight": (d, n_verts), "vertice_map.bias": (d,),
        "vertice_map_r.weight": (n_verts, d), "vertice_map_r.bias": (n_verts,),
        "obj_vector.weight": (d, n_onehot),
    }
    p = "transformer_decoder.layers.0."
    for att in ("self_attn", "multihead_attn"):
        s[p + att + ".in_proj_weight"] = (3 * d, d)
        s[p + att + ".in_proj_bias"] = (3 * d,)
        s[p + att + ".out_proj.weight"] = (d, d)
        s[p + att + ".out_proj.bias"] = (d,)
    s[p + "linear1.weight"] = (2 * d, d); s[p + "linear1.bias"] = (2 * d,)
    s[p + "linear2.weight"] = (d, 2 * d); s[p + "linear2.bias"] = (d,)
    for n in ("norm1", "norm2", "norm3"):
        s[p + n + ".weight"] = (d,); s[p + n + ".bias"] = (d,)
    return s


def wav2vec2_shapes(hidden: int = 768, layers: int = 12, ffn: int = 3072, conv_dim: int = 512,
                    kernels=(10, 3, 3, 3, 3, 2, 2), pos_k: int = 128, pos_groups: int = 16) -> Dict[str, Tuple[int, ...]]:
    """Keys/shapes of HF Wav2Vec2Model.state_dict() (transformers 5.15 naming, wav2vec2-base config)."""
    s = {"masked_spec_embed": (hidden,)}
    cin = 1
    for i, k in enumerate(kernels):
        s[f"feature_extractor.conv_layers.{i}.conv.weight"] = (conv_dim, cin, k)
        if i == 0:
            s["feature_extractor.conv_layers.0.layer_norm.weight"] = (conv_dim,)
            s["feature_extractor.conv_layers.0.layer_norm.bias"] = (conv_dim,)
        cin = conv_dim
    s["feature_projection.layer_norm.weight"] = (conv_dim,)
    s["feature_projection.layer_norm.bias"] = (conv_dim,)
    s["feature_projection.projection.weight"] = (hidden, conv_dim)
    s["feature_projection.projection.bias"] = (hidden,)
    s["encoder.pos_conv_embed.conv.bias"] = (hidden,)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, pos_k)
    s["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (hidden, hidden // pos_groups, pos_k)
    s["encoder.layer_norm.weight"] = (hidden,)
    s["encoder.layer_norm.bias"] = (hidden,)
    for l in range(layers):
        p = f"encoder.layers.{l}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"attention.{nm}.weight"] = (hidden, hidden)
            s[p + f"attention.{nm}.bias"] = (hidden,)
        s[p + "layer_norm.weight"] = (hidden,); s[p + "layer_norm.bias"] = (hidden,)
        s[p + "feed_forward.intermediate_dense.weight"] = (ffn, hidden)
        s[p + "feed_forward.intermediate_dense.bias"] = (ffn,)
        s[p + "feed_forward.output_dense.weight"] = (hidden, ffn)
        s[p + "feed_forward.output_dense.bias"] = (hidden,)
        s[p + "final_layer_norm.weight"] = (hidden,); s[p + "final_layer_norm.bias"] = (hidden,)
    return s
