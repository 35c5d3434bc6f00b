"""Train-mode stochastic regularisers of the wav2vec2 encoder (host side): which masks a step needs, how they are drawn, how
recorded masks are injected.

What the reference does in train mode (HF ``Wav2Vec2Model`` under ``/root/reference/src/model/wav2vec.py:87-189`` with the
configuration of ``facebook/wav2vec2-base-960h``): dropout 0.1 after the feature projection, after the encoder's first
LayerNorm, on the attention probabilities, on the attention block's output, on GELU(intermediate_dense) and on the
feed-forward output of each of the 12 layers; LayerDrop 0.1 (one uniform draw per layer, the layer is skipped when it falls
below); SpecAugment along time (``_compute_mask_indices``, wav2vec.py:25-72: ``mask_time_prob`` 0.05, spans of 10 frames, at
least 2 spans - the masked frames are replaced by the learned ``masked_spec_embed``, wav2vec.py:149-162).

The kernels consume bit-packed keep masks (csrc/dropout.hip).  ``draw_encoder_masks`` fills them on the device from one
64-bit step seed (Philox4x32-10: a mask is a pure function of (seed, site, element)); ``EncoderMasks.from_recorded`` packs
boolean masks recorded elsewhere - the parity tests record every mask the reference drew in one train-mode forward and feed
exactly those (tests/golden/gen_golden.py, G13), which pins the train-mode function although it is stochastic.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch


def pack_keep_bits(mask: torch.Tensor) -> torch.Tensor:
    """bool (R, C) -> int32 (R, ceil(C / 32)): bit b of word w <-> column 32 w + b (layout of a2f_keep_bits)."""
    R, C = mask.shape
    wpr = (C + 31) // 32
    m = torch.zeros(R, wpr * 32, dtype=torch.int64, device=mask.device)
    m[:, :C] = mask.to(torch.int64)
    w = (m.view(R, wpr, 32) << torch.arange(32, dtype=torch.int64, device=mask.device)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def unpack_keep_bits(bits: torch.Tensor, C: int) -> torch.Tensor:
    """int32 (R, wpr) -> bool (R, C)."""
    w = bits.to(torch.int64) & 0xFFFFFFFF
    sh = torch.arange(32, dtype=torch.int64, device=bits.device)
    return (((w.unsqueeze(-1) >> sh) & 1) != 0).reshape(bits.shape[0], -1)[:, :C]


def attn_keep_words(T: int) -> int:
    return 2 * ((T + 63) // 64)


def pack_keep_bits_attn(mask: torch.Tensor):
    """bool (SH, T, T) [sh][query][key] -> (query-major int32 (SH*T, TW), key-major int32 (SH, TW, 32 TW)) as a2f_keep_bits_attn
    lays them out (TW = 2 ceil(T / 64); key-major word (qw, key): bit b <-> query 32 qw + b)."""
    SH, T, _ = mask.shape
    TW = attn_keep_words(T)
    full = torch.zeros(SH, 32 * TW, 32 * TW, dtype=torch.bool, device=mask.device)
    full[:, :T, :T] = mask
    qmaj = pack_keep_bits(full[:, :T].reshape(SH * T, 32 * TW))
    kmaj = pack_keep_bits(full.transpose(1, 2).reshape(SH * 32 * TW, 32 * TW)).view(SH, 32 * TW, TW).transpose(1, 2).contiguous()
    return qmaj, kmaj


def compute_mask_indices(shape, mask_prob: float, mask_length: int, min_masks: int = 0, rng=None) -> np.ndarray:
    """SpecAugment span mask, restated from /root/reference/src/model/wav2vec.py:25-72 (no attention_mask: FaceFormer never passes
    one).  Consumes the random stream exactly as the reference does - one ``rand()``, one ``choice`` of span starts per row, one
    ``choice`` per row that is thinned to the shortest row - so the same numpy seed gives the same mask (pinned by fixture G13).
    ``rng``: a ``np.random.RandomState``; default = numpy's global state, which is what the reference draws from."""
    rng = np.random if rng is None else rng
    rows, size = shape
    n_spans = max(min_masks, int(mask_prob * size / float(mask_length) + rng.rand()))
    picked = []
    for _ in range(rows):
        lengths = np.full(n_spans, mask_length)
        if lengths.sum() == 0:
            lengths[0] = min(mask_length, size - 1)
        shortest = lengths.min()
        if size - shortest <= n_spans:
            shortest = size - n_spans - 1
        starts = rng.choice(size - shortest, n_spans, replace=False)
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1) if n_spans else np.zeros(0, dtype=np.int64)
        picked.append(np.unique(idx[idx < size]))
    keep_n = min(len(p) for p in picked)
    mask = np.zeros((rows, size), dtype=bool)
    for r, idx in enumerate(picked):
        if len(idx) > keep_n:
            idx = rng.choice(idx, keep_n, replace=False)
        mask[r, idx] = True
    return mask


class EncoderMasks:
    """The masks of one train-mode forward of the encoder: ``featproj`` / ``enc`` = (bits, scale) or None; ``layers`` = one dict per
    Transformer layer as EncoderLayerFn takes it; ``layer_skip`` = LayerDrop decisions; ``spec`` = bool (S, T) rows replaced by
    ``masked_spec_embed`` (or None)."""

    def __init__(self):
        self.featproj = None
        self.enc = None
        self.layers: List[Dict] = []
        self.layer_skip: List[bool] = []
        self.spec: Optional[torch.Tensor] = None
        self.layer_keep_dev: Optional[torch.Tensor] = None     # graph-captured steps: bool (L,) on the device, every layer runs and this selects

    @staticmethod
    def from_recorded(rec: Dict, device, p_feat=0.1, p_hidden=0.1, p_attn=0.1, p_act=0.1) -> "EncoderMasks":
        """rec: {"featproj": bool (M, d), "enc": bool (M, d), "layer{l}.attn": bool (S*H, T, T), "layer{l}.hid1" / ".act" / ".hid2": bool
        (M, ·), "layer_skip": bool (L,), "spec": bool (S, T)} (numpy or torch); missing entries = no dropout at that site."""
        t = lambda a: torch.as_tensor(np.asarray(a)).to(device)
        m = EncoderMasks()
        if "featproj" in rec:
            m.featproj = (pack_keep_bits(t(rec["featproj"])), 1.0 / (1.0 - p_feat))
        if "enc" in rec:
            m.enc = (pack_keep_bits(t(rec["enc"])), 1.0 / (1.0 - p_hidden))
        skip = [bool(x) for x in np.asarray(rec["layer_skip"]).reshape(-1)] if "layer_skip" in rec else []
        L = len(skip) if skip else 1 + max([int(k.split(".")[0][5:]) for k in rec if k.startswith("layer") and "." in k], default=-1)
        for l in range(L):
            d = {}
            if f"layer{l}.attn" in rec:
                q, k = pack_keep_bits_attn(t(rec[f"layer{l}.attn"]))
                d["attn"] = (q, k, 1.0 / (1.0 - p_attn))
            for name, p in (("hid1", p_hidden), ("act", p_act), ("hid2", p_hidden)):
                if f"layer{l}.{name}" in rec:
                    d[name] = (pack_keep_bits(t(rec[f"layer{l}.{name}"])), 1.0 / (1.0 - p))
            m.layers.append(d)
        m.layer_skip = skip or [False] * L
        if "spec" in rec:
            m.spec = t(rec["spec"]).bool()
        return m


_PINNED = {}


def _to_device_async(a: np.ndarray, device) -> torch.Tensor:
    """Host array -> device tensor WITHOUT blocking the host.  A plain ``.to(device)`` of pageable memory is a synchronous, stream-ordered
    copy: the host waits until every kernel already queued on the stream has run - once per train step, i.e. the host could never issue
    a step ahead of the GPU (measured on the one-sequence FaceFormer step: 9 of 20 ms of host time sat in this copy).  Here the bytes
    go through a small ring of pinned staging buffers and a non-blocking copy; a slot is reused only after its last copy has completed."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.from_numpy(a).to(dev)
    key = (a.shape, a.dtype.str, dev.index)
    ring = _PINNED.setdefault(key, {"slots": [], "next": 0})
    if len(ring["slots"]) < 4:
        ring["slots"].append([torch.empty(a.shape, dtype=torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype).pin_memory(), None])
        slot = ring["slots"][-1]
    else:
        slot = ring["slots"][ring["next"] % 4]
        ring["next"] += 1
        if slot[1] is not None:
            slot[1].synchronize()
    slot[0].copy_(torch.from_numpy(a))
    out = slot[0].to(dev, non_blocking=True)
    slot[1] = torch.cuda.current_stream(dev).record_event()
    return out


class StepRandomness:
    """The per-step random state of a FaceFormer train step as DEVICE-RESIDENT tensors, for hipGraph-captured steps: a captured launch cannot take a fresh
    seed by value, and a host-side LayerDrop branch or SpecAugment mask cannot change between replays.  ``refresh()`` draws on the host exactly what the eager
    step draws - one 62-bit encoder seed from torch's CPU generator, LayerDrop and SpecAugment from the numpy stream seeded with it, one decoder seed - and copies
    it into the static tensors (pinned staging, non-blocking); the captured kernels read ``seed`` / ``dec_seed`` at run time (a2f_keep_bits_dev), every layer is
    launched and ``layer_keep[l]`` selects its output or its input (a dropped layer's gradients come out as exact zeros: what FlatAdam.collect_grads fills in
    for a layer autograd never visited), ``spec`` is applied as a select.  Same seed -> same masks, decisions and results as the eager step."""

    def __init__(self, device, n_layers: int, S: int, T: int):
        self.device, self.n_layers, self.S, self.T = torch.device(device), n_layers, S, T
        self.seed = torch.zeros(1, dtype=torch.int64, device=device)
        self.dec_seed = torch.zeros(1, dtype=torch.int64, device=device)
        self.layer_keep = torch.ones(n_layers, dtype=torch.bool, device=device)
        self.spec = torch.zeros(S, T, dtype=torch.bool, device=device)
        self._stage = None
        self.last = None                      # the host-side values of the last refresh (tests)

    def refresh(self, cfg, train_encoder: bool = True, train_decoder: bool = True):
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train_encoder else 0
        rng = np.random.RandomState(seed & 0x7FFFFFFF)
        keep = np.ones(self.n_layers, dtype=bool)
        spec = np.zeros((self.S, self.T), dtype=bool)
        if train_encoder:
            for l in range(self.n_layers):
                keep[l] = not (rng.rand() < cfg.layerdrop)
            if cfg.apply_spec_augment and cfg.mask_time_prob > 0.0 and self.T > cfg.mask_time_length:
                spec = compute_mask_indices((self.S, self.T), cfg.mask_time_prob, cfg.mask_time_length, min_masks=2, rng=rng)
        dec_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train_decoder else 0
        self.last = {"seed": seed, "dec_seed": dec_seed, "layer_keep": keep.copy(), "spec": spec.copy()}
        if self.device.type != "cuda":
            self.seed[0], self.dec_seed[0] = seed, dec_seed
            self.layer_keep.copy_(torch.from_numpy(keep))
            self.spec.copy_(torch.from_numpy(spec))
            return
        if self._stage is None:
            self._stage = {"seeds": torch.zeros(2, dtype=torch.int64).pin_memory(), "keep": torch.zeros(self.n_layers, dtype=torch.bool).pin_memory(),
                           "spec": torch.zeros(self.S, self.T, dtype=torch.bool).pin_memory(), "ev": None}
        st = self._stage
        if st["ev"] is not None:
            st["ev"].synchronize()            # the previous step's copies have left the staging buffers
        st["seeds"][0], st["seeds"][1] = seed, dec_seed
        st["keep"].copy_(torch.from_numpy(keep))
        st["spec"].copy_(torch.from_numpy(spec))
        self.seed.copy_(st["seeds"][0:1], non_blocking=True)
        self.dec_seed.copy_(st["seeds"][1:2], non_blocking=True)
        self.layer_keep.copy_(st["keep"], non_blocking=True)
        self.spec.copy_(st["spec"], non_blocking=True)
        st["ev"] = torch.cuda.current_stream(self.device).record_event()


def draw_encoder_masks(be, S: int, T: int, n_layers: int, heads: int, hidden: int, ffn: int, cfg, seed, device, static: "StepRandomness" = None) -> EncoderMasks:
    """The masks of one step, drawn on the device from ``seed`` (sites: 0 feature projection, 1 encoder, 2 + 4 l + {0 attention,
    1 attention output, 2 activation, 3 feed-forward output}); LayerDrop and SpecAugment draw from a numpy stream seeded with it.
    ``static`` (graph-captured steps): seed, LayerDrop decisions and SpecAugment rows come from its device tensors - masks for every layer."""
    M = S * T
    m = EncoderMasks()
    if static is not None:
        seed = static.seed
        m.layer_keep_dev = static.layer_keep
    words = lambda c: torch.empty(M, (c + 31) // 32, dtype=torch.int32, device=device)

    def site(idx, cols, p):
        if p <= 0.0:
            return None
        b = words(cols)
        be.keep_bits(seed, idx, M, cols, p, b)
        return (b, 1.0 / (1.0 - p))

    m.featproj = site(0, hidden, cfg.feat_proj_dropout)
    m.enc = site(1, hidden, cfg.hidden_dropout)
    rng = np.random.RandomState(seed & 0x7FFFFFFF) if static is None else None
    TW = attn_keep_words(T)
    for l in range(n_layers):
        d = {}
        m.layer_skip.append(bool(rng.rand() < cfg.layerdrop) if static is None else False)   # HF Wav2Vec2Encoder: skip when the draw is below layerdrop
        if m.layer_skip[-1]:
            m.layers.append(d)                                             # a dropped layer needs no masks
            continue
        if cfg.attention_dropout > 0.0:
            q = torch.empty(S * heads * T, TW, dtype=torch.int32, device=device)
            k = torch.empty(S * heads, TW, 32 * TW, dtype=torch.int32, device=device)
            be.keep_bits_attn(seed, 2 + 4 * l, S * heads, T, cfg.attention_dropout, q, k)
            d["attn"] = (q, k, 1.0 / (1.0 - cfg.attention_dropout))
        for j, (name, cols, p) in enumerate((("hid1", hidden, cfg.hidden_dropout), ("act", ffn, cfg.activation_dropout), ("hid2", hidden, cfg.hidden_dropout))):
            s_ = site(2 + 4 * l + 1 + j, cols, p)
            if s_ is not None:
                d[name] = s_
        m.layers.append(d)
    if static is not None:
        m.spec = static.spec
    elif cfg.apply_spec_augment and cfg.mask_time_prob > 0.0 and T > cfg.mask_time_length:
        m.spec = _to_device_async(compute_mask_indices((S, T), cfg.mask_time_prob, cfg.mask_time_length, min_masks=2, rng=rng), device)
    return m


# ---------------------------------------------------------------------------------------------------------------------------
# FaceFormer decoder / PPE dropout (faceformer.py:73,88; nn.TransformerDecoderLayer(dropout = 0.1 default) at faceformer.py:121-127)
# ---------------------------------------------------------------------------------------------------------------------------
class DecoderMasks:
    """Per-row keep masks of the decoder recurrence (one mask per frame row - see csrc/ff_decoder.hip for why not per loop step):
    ``row`` int32 (S*T, 10): columns 0..63 PPE output, 64..127 dropout1, 128..255 feed-forward activation, 256..319 dropout3;
    ``attn`` int32 (S*4*T, TW): self-attention probabilities of (sequence, head, frame), bit = key; ``cross_v`` int32 (S*T, 2): the
    cross-attention probability dropout - the diagonal memory mask (faceformer.py:58-66) leaves ONE key per frame, so that dropout
    keeps or drops a whole head's value vector: one decision per (frame, head) spread over the head's 16 columns; ``cross_out`` int32
    (S*T, 2): dropout2.  ``scale`` = 1 / (1 - p)."""

    def __init__(self, row, attn, cross_v, cross_out, scale):
        self.row, self.attn, self.cross_v, self.cross_out, self.scale = row, attn, cross_v, cross_out, float(scale)

    @staticmethod
    def from_recorded(rec: Dict, device, p: float = 0.1) -> "DecoderMasks":
        """rec: bool arrays "ppe" (S*T, 64), "d1" (S*T, 64), "act" (S*T, 128), "d3" (S*T, 64), "attn" (S*4, T, T) [head-major rows s*4 + h],
        "cross_head" (S*T, 4), "d2" (S*T, 64)."""
        t = lambda a: torch.as_tensor(np.asarray(a)).to(device).bool()
        row = pack_keep_bits(torch.cat([t(rec["ppe"]), t(rec["d1"]), t(rec["act"]), t(rec["d3"])], dim=1))
        a = t(rec["attn"])
        SH, T, _ = a.shape
        attn = pack_keep_bits(torch.nn.functional.pad(a.reshape(SH * T, T), (0, 32 * attn_keep_words(T) - T)))
        cross_v = pack_keep_bits(t(rec["cross_head"]).repeat_interleave(16, dim=1))
        return DecoderMasks(row, attn, cross_v, pack_keep_bits(t(rec["d2"])), 1.0 / (1.0 - p))


def draw_decoder_masks(be, S: int, T: int, p: float, seed, device) -> DecoderMasks:
    """Masks of one step from ``seed`` (sites 1000 row block, 1001 self-attention, 1002 cross-attention heads, 1003 dropout2)."""
    M = S * T
    row = torch.empty(M, 10, dtype=torch.int32, device=device)
    be.keep_bits(seed, 1000, M, 320, p, row)
    attn = torch.empty(S * 4 * T, attn_keep_words(T), dtype=torch.int32, device=device)
    be.keep_bits(seed, 1001, S * 4 * T, T, p, attn)
    heads = torch.empty(M, 1, dtype=torch.int32, device=device)
    be.keep_bits(seed, 1002, M, 4, p, heads)
    cross_v = pack_keep_bits(unpack_keep_bits(heads, 4).repeat_interleave(16, dim=1))
    cross_out = torch.empty(M, 2, dtype=torch.int32, device=device)
    be.keep_bits(seed, 1003, M, 64, p, cross_out)
    return DecoderMasks(row, attn, cross_v, cross_out, 1.0 / (1.0 - p))
