"""FaceFormer decoder, CPU restatement (test infrastructure, see oracle/__init__.py).

Restates /root/reference/src/model/faceformer.py:
* ``biased_mask_table``     <- init_biased_mask, faceformer.py:22-54 (table form, as the reference builds it)
* ``biased_mask_closed``    <- same values in closed form (SURVEY §3.3 identity 1)
* ``enc_dec_mask_vocaset``  <- enc_dec_mask, faceformer.py:58-66 ("vocaset" branch)
* ``ppe_table``             <- PeriodicPositionalEncoding, faceformer.py:70-88 (dropout off)
* ``decoder_layer``         <- nn.TransformerDecoderLayer(d=64, nhead=4, ff=128, batch_first, post-norm, relu)
                               as instantiated at faceformer.py:121-127 (torch's own maths, written out)
* ``decode_reference_loop`` <- Faceformer.forward's autoregressive loop, faceformer.py:147-188 (O(T^3) as written)
* ``decode_incremental``    <- the same function evaluated with a K/V cache and the collapsed 64x64 feedback
                               (what the HIP recurrence kernel implements; equal up to fp32 rounding)

All functions are eval-mode (dropout off): the reference is stochastic in train mode.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

LAYER = "transformer_decoder.layers.0."


# ----------------------------------------------------------------------------- masks / PPE
def alibi_slopes(n_head: int):
    """faceformer.py:23-37 (power-of-two and general branch)."""
    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(n_head).is_integer():
        return pow2(n_head)
    c = 2 ** math.floor(math.log2(n_head))
    return pow2(c) + alibi_slopes(2 * c)[0::2][: n_head - c]


def biased_mask_table(n_head: int, max_seq_len: int, period: int) -> torch.Tensor:
    """(n_head, L, L): -slope_h * floor((i-j)/period) on/below the diagonal, -inf above.  faceformer.py:39-54."""
    slopes = torch.tensor(alibi_slopes(n_head), dtype=torch.float32)
    steps = torch.arange(0, max_seq_len, period).unsqueeze(1).repeat(1, period).view(-1) // period
    steps = -torch.flip(steps, dims=[0])
    tab = torch.zeros(max_seq_len, max_seq_len)
    for i in range(max_seq_len):
        tab[i, : i + 1] = steps[-(i + 1):]
    tab = slopes.view(-1, 1, 1) * tab.unsqueeze(0)
    causal = torch.zeros(max_seq_len, max_seq_len).masked_fill(
        torch.triu(torch.ones(max_seq_len, max_seq_len, dtype=torch.bool), diagonal=1), float("-inf"))
    return causal.unsqueeze(0) + tab


def biased_mask_closed(n_head: int, T: int, period: int) -> torch.Tensor:
    i = torch.arange(T).view(T, 1)
    j = torch.arange(T).view(1, T)
    slopes = torch.tensor(alibi_slopes(n_head), dtype=torch.float32).view(-1, 1, 1)
    steps = torch.div(i - j, period, rounding_mode="floor").to(torch.float32)
    m = -(slopes * steps.unsqueeze(0))
    return m.masked_fill((j > i).unsqueeze(0), float("-inf"))


def enc_dec_mask_vocaset(T: int, S: int) -> torch.Tensor:
    """Bool (T,S), True = masked; only (i,i) is visible.  faceformer.py:58-66."""
    m = torch.ones(T, S, dtype=torch.bool)
    idx = torch.arange(min(T, S))
    m[idx, idx] = False
    return m


def ppe_table(d_model: int, period: int, length: int) -> torch.Tensor:
    """(length, d_model) sinusoid evaluated at p mod period.  faceformer.py:74-83."""
    pos = torch.arange(0, period, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe = torch.zeros(period, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    reps = length // period + 1
    return pe.repeat(reps, 1)[:length]


# ----------------------------------------------------------------------------- decoder layer (torch maths written out)
def _split_heads(x: torch.Tensor, nhead: int) -> torch.Tensor:      # (L,d) -> (h,L,hd)
    L, d = x.shape
    return x.view(L, nhead, d // nhead).transpose(0, 1)


def _attention(P, pre: str, xq: torch.Tensor, xkv: torch.Tensor, nhead: int, add_mask=None, bool_mask=None) -> torch.Tensor:
    """nn.MultiheadAttention forward (eval, need_weights irrelevant) on unbatched (L,d)/(S,d) inputs."""
    d = xq.shape[-1]
    W, b = P[pre + ".in_proj_weight"], P[pre + ".in_proj_bias"]
    q = _split_heads(F.linear(xq, W[:d], b[:d]), nhead)
    k = _split_heads(F.linear(xkv, W[d:2 * d], b[d:2 * d]), nhead)
    v = _split_heads(F.linear(xkv, W[2 * d:], b[2 * d:]), nhead)
    s = torch.matmul(q * (1.0 / math.sqrt(d // nhead)), k.transpose(1, 2))     # (h,L,S)
    if add_mask is not None:
        s = s + add_mask
    if bool_mask is not None:
        s = s.masked_fill(bool_mask.unsqueeze(0), float("-inf"))
    a = torch.softmax(s, dim=-1)
    ctx = torch.matmul(a, v).transpose(0, 1).reshape(xq.shape[0], d)
    return F.linear(ctx, P[pre + ".out_proj.weight"], P[pre + ".out_proj.bias"])


def _ln(P, key, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[key + ".weight"], P[key + ".bias"], eps)


def decoder_layer(P, tgt: torch.Tensor, memory: torch.Tensor, tgt_mask: torch.Tensor, memory_mask: torch.Tensor, nhead: int = 4) -> torch.Tensor:
    """Post-norm decoder layer on (L,d) target and (S,d) memory."""
    x = _ln(P, LAYER + "norm1", tgt + _attention(P, LAYER + "self_attn", tgt, tgt, nhead, add_mask=tgt_mask))
    x = _ln(P, LAYER + "norm2", x + _attention(P, LAYER + "multihead_attn", x, memory, nhead, bool_mask=memory_mask))
    h = F.linear(F.relu(F.linear(x, P[LAYER + "linear1.weight"], P[LAYER + "linear1.bias"])),
                 P[LAYER + "linear2.weight"], P[LAYER + "linear2.bias"])
    return _ln(P, LAYER + "norm3", x + h)


# ----------------------------------------------------------------------------- the autoregressive loop, as written
def decode_reference_loop(P: Dict[str, torch.Tensor], enc: torch.Tensor, one_hot: torch.Tensor, template: torch.Tensor,
                          period: int = 60, nhead: int = 4, max_seq_len: int = 600) -> torch.Tensor:
    """enc: encoder last_hidden_state (1,T,768); one_hot (1,n_onehot); template (1,V,3) -> (1,T,V,3).

    Literal restatement of faceformer.py:147-188: the whole decoder and the 64->15069 map are re-run on the
    growing prefix each step; only the last step's output is returned.
    """
    T = enc.shape[1]
    d = P["obj_vector.weight"].shape[0]
    mem = F.linear(enc[0], P["audio_feature_map.weight"], P["audio_feature_map.bias"])      # (T,64)  :152
    style = F.linear(one_hot, P["obj_vector.weight"])                                       # (1,64)  :148
    table = biased_mask_table(nhead, max(max_seq_len, T), period) if T <= max_seq_len else biased_mask_closed(nhead, T, period)
    pe = ppe_table(d, period, T + period)
    emb = style.clone()                                                                     # (1,64)  :156
    out = None
    for i in range(T):                                                                      # :154
        L = emb.shape[0]
        tgt = emb + pe[:L]                                                                  # PPE, :157-160
        y = decoder_layer(P, tgt, mem, table[:, :L, :L], enc_dec_mask_vocaset(L, T), nhead)  # :162-179
        out = F.linear(y, P["vertice_map_r.weight"], P["vertice_map_r.bias"])               # (L,V*3) :181
        nxt = F.linear(out[-1:], P["vertice_map.weight"], P["vertice_map.bias"]) + style    # :183-184
        emb = torch.cat((emb, nxt), 0)                                                      # :185
    out = out + template.reshape(1, -1)                                                     # :187
    return out.view(1, T, -1, 3)                                                            # :188


# ----------------------------------------------------------------------------- incremental (K/V cache) evaluation
def collapsed_feedback(P) -> Tuple[torch.Tensor, torch.Tensor]:
    """vertice_map(vertice_map_r(h)) == A h + c with A = Wm Wr (64x64), c = Wm br + bm.  SURVEY §3.3 identity 5."""
    Wm, bm = P["vertice_map.weight"], P["vertice_map.bias"]
    Wr, br = P["vertice_map_r.weight"], P["vertice_map_r.bias"]
    return Wm @ Wr, Wm @ br + bm


def cross_attention_diagonal(P, mem: torch.Tensor) -> torch.Tensor:
    """With only (i,i) visible the cross-attention softmax has one element:
    cross(q, mem)[i] = out_proj(v_proj(mem[i])).  SURVEY §3.3 identity 3."""
    pre = LAYER + "multihead_attn"
    d = mem.shape[-1]
    W, b = P[pre + ".in_proj_weight"], P[pre + ".in_proj_bias"]
    v = F.linear(mem, W[2 * d:], b[2 * d:])
    return F.linear(v, P[pre + ".out_proj.weight"], P[pre + ".out_proj.bias"])


def decode_incremental_hidden(P, enc: torch.Tensor, one_hot: torch.Tensor, period: int = 60, nhead: int = 4) -> torch.Tensor:
    """Returns the decoder output rows y (T,64); pred = vertice_map_r(y) + template."""
    T = enc.shape[1]
    d = P["obj_vector.weight"].shape[0]
    hd = d // nhead
    mem = F.linear(enc[0], P["audio_feature_map.weight"], P["audio_feature_map.bias"])
    cross = cross_attention_diagonal(P, mem)                                               # (T,64)
    style = F.linear(one_hot, P["obj_vector.weight"])[0]                                   # (64,)
    A, c = collapsed_feedback(P)
    pe = ppe_table(d, period, T + period)
    slopes = torch.tensor(alibi_slopes(nhead), dtype=torch.float32)
    pre = LAYER + "self_attn"
    W, b = P[pre + ".in_proj_weight"], P[pre + ".in_proj_bias"]
    ks, vs, ys = [], [], []          # K/V cache kept as lists so autograd can differentiate through it
    emb = style
    for i in range(T):
        x = emb + pe[i]
        qkv = F.linear(x, W, b)
        q = qkv[:d]
        ks.append(qkv[d:2 * d])
        vs.append(qkv[2 * d:])
        qh = q.view(nhead, hd) * (1.0 / math.sqrt(hd))
        kh = torch.stack(ks, 0).view(i + 1, nhead, hd)
        s = torch.einsum("hd,jhd->hj", qh, kh)
        steps = torch.div(i - torch.arange(i + 1), period, rounding_mode="floor").to(torch.float32)
        s = s - slopes.view(-1, 1) * steps.view(1, -1)
        a = torch.softmax(s, dim=-1)
        ctx = torch.einsum("hj,jhd->hd", a, torch.stack(vs, 0).view(i + 1, nhead, hd)).reshape(d)
        y = _ln(P, LAYER + "norm1", x + F.linear(ctx, P[pre + ".out_proj.weight"], P[pre + ".out_proj.bias"]))
        y = _ln(P, LAYER + "norm2", y + cross[i])
        h = F.linear(F.relu(F.linear(y, P[LAYER + "linear1.weight"], P[LAYER + "linear1.bias"])),
                     P[LAYER + "linear2.weight"], P[LAYER + "linear2.bias"])
        y = _ln(P, LAYER + "norm3", y + h)
        ys.append(y)
        emb = A @ y + c + style
    return torch.stack(ys, 0)


def decode_incremental(P, enc, one_hot, template, period: int = 60, nhead: int = 4) -> torch.Tensor:
    y = decode_incremental_hidden(P, enc, one_hot, period, nhead)
    out = F.linear(y, P["vertice_map_r.weight"], P["vertice_map_r.bias"]) + template.reshape(1, -1)
    return out.view(1, y.shape[0], -1, 3)
