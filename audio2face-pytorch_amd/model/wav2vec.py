"""wav2vec2-base audio encoder as FaceFormer uses it (reference: /root/reference/src/model/wav2vec.py:76-189, which
subclasses HF ``transformers.Wav2Vec2Model``; this is a self-contained module with HF-compatible state-dict keys, so
``facebook/wav2vec2-base-960h`` weights load with ``load_state_dict``).

Arithmetic: third-party (HF) architecture.  ``forward`` runs the hand-written path and nothing else (``encoder_ops``: every
contraction on the bf16-MFMA GEMM family ``a2f_mm`` - feature-encoder convolutions 1..6 as implicit GEMMs, feature projection,
grouped positional convolution, the 12 Transformer layers - MFMA flash attention, HIP LayerNorm / GELU' / bias-gradient
kernels, forward and backward).  There is no torch-ops forward in the product: off-device tensors fail in ``ops._chk``; the torch-ops
restatement the CPU suite compares against lives in ``tests/torch_ops_encoder.py``.  ``forward(input_values, dataset, frame_num=...)`` keeps the reference wrapper's signature and
returns an object with ``.last_hidden_state`` (1,T,768); the vocaset branch interpolates the 49 Hz conv features to
``frame_num`` frames with align_corners=True (wav2vec.py:76-84,125-128).

Train mode (``.train()``, as Lightning puts the reference's module): the stochastic regularisers of the reference's encoder are
applied - dropout at the six kinds of sites of HF's modules, LayerDrop, SpecAugment along time (wav2vec.py:149-162) - with the
probabilities of ``facebook/wav2vec2-base-960h`` (``self.config``).  Masks are drawn per forward from torch's CPU generator
(``dropout.draw_encoder_masks``) or injected (``self.injected_masks``: how the parity test feeds the masks recorded from the
reference, fixture G13).  ``self.train_regularisers = False`` keeps the deterministic function in train mode.
"""
from __future__ import annotations

import types

import torch
import torch.nn as nn

CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)


class _ConvLayer(nn.Module):
    def __init__(self, cin, cout, k, group_norm):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, bias=False)
        if group_norm:
            self.layer_norm = nn.GroupNorm(cout, cout, affine=True)


class _FeatureExtractor(nn.Module):
    def __init__(self, dim=512):
        super().__init__()
        self.conv_layers = nn.ModuleList(_ConvLayer(1 if i == 0 else dim, dim, k, i == 0) for i, k in enumerate(CONV_KERNEL))


class _FeatureProjection(nn.Module):
    def __init__(self, cin=512, hidden=768):
        super().__init__()
        self.layer_norm = nn.LayerNorm(cin, eps=1e-5)
        self.projection = nn.Linear(cin, hidden)


class _WeightNormDim2Fn(torch.autograd.Function):
    """w = v * g / ||v||, the norm over dims (0, 1) of a (Co, Ci/groups, k) weight (torch._weight_norm with dim = 2), as a handful of streaming torch ops:
    torch's own fused kernels for a non-leading `dim` take 0.46 + 0.49 ms per call on the 4.7 M-element positional-convolution weight (rocprofv3, round 4)."""

    @staticmethod
    def forward(ctx, g, v):
        n = torch.sqrt((v * v).sum((0, 1), keepdim=True))
        s = g / n
        ctx.save_for_backward(v, s, n)
        return v * s

    @staticmethod
    def backward(ctx, dw):
        v, s, n = ctx.saved_tensors
        t = (dw * v).sum((0, 1), keepdim=True)            # d/ds
        dg = t / n
        dv = dw * s - v * (t * s / (n * n))
        return dg, dv


class _WeightNormDim2(nn.Module):
    """Parametrization with torch.nn.utils.parametrizations.weight_norm's state-dict layout (original0 = g (1, 1, k), original1 = v)."""

    def forward(self, weight_g, weight_v):
        return _WeightNormDim2Fn.apply(weight_g, weight_v)

    def right_inverse(self, weight):
        return torch.norm_except_dim(weight, 2, 2), weight


class _PosConv(nn.Module):
    def __init__(self, hidden=768, k=128, groups=16):
        super().__init__()
        self.conv = nn.Conv1d(hidden, hidden, k, padding=k // 2, groups=groups)
        nn.utils.parametrize.register_parametrization(self.conv, "weight", _WeightNormDim2())
        self.conv._register_load_state_dict_pre_hook(_PosConv._legacy_weight_norm_keys)
        self.k = k

    @staticmethod
    def _legacy_weight_norm_keys(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """What torch.nn.utils.parametrizations.weight_norm registers (_weight_norm_compat_hook): checkpoints written through the old nn.utils.weight_norm - the
        published facebook/wav2vec2-base-960h weights and Lightning checkpoints of older torch / transformers - hold `conv.weight_g` / `conv.weight_v`."""
        for old, new in (("weight_g", "parametrizations.weight.original0"), ("weight_v", "parametrizations.weight.original1")):
            if prefix + old in state_dict and prefix + new not in state_dict:
                state_dict[prefix + new] = state_dict.pop(prefix + old)


class _Attention(nn.Module):
    def __init__(self, hidden=768):
        super().__init__()
        self.k_proj = nn.Linear(hidden, hidden)
        self.v_proj = nn.Linear(hidden, hidden)
        self.q_proj = nn.Linear(hidden, hidden)
        self.out_proj = nn.Linear(hidden, hidden)


class _FeedForward(nn.Module):
    def __init__(self, hidden=768, ffn=3072):
        super().__init__()
        self.intermediate_dense = nn.Linear(hidden, ffn)
        self.output_dense = nn.Linear(ffn, hidden)


class _EncoderLayer(nn.Module):
    def __init__(self, hidden=768, ffn=3072, heads=12):
        super().__init__()
        self.attention = _Attention(hidden)
        self.layer_norm = nn.LayerNorm(hidden, eps=1e-5)
        self.feed_forward = _FeedForward(hidden, ffn)
        self.final_layer_norm = nn.LayerNorm(hidden, eps=1e-5)
        self.heads = heads


class _Encoder(nn.Module):
    def __init__(self, hidden=768, layers=12, ffn=3072, heads=12):
        super().__init__()
        self.pos_conv_embed = _PosConv(hidden)
        self.layer_norm = nn.LayerNorm(hidden, eps=1e-5)
        self.layers = nn.ModuleList(_EncoderLayer(hidden, ffn, heads) for _ in range(layers))


def linear_interpolation(features, output_len):
    """(B,L,C) -> (B,output_len,C), align_corners=True   (wav2vec.py:76-84): csrc/interp.hip, gather-form backward, no atomics."""
    from ..functional import LinearInterpFn
    return LinearInterpFn.apply(features.float(), int(output_len)).to(features.dtype)


def normalize_waveform(audio):
    """Wav2Vec2FeatureExtractor(do_normalize=True): per-utterance zero mean / unit variance (faceformer.py:142-144)."""
    x = audio.to(torch.float32)
    m = x.mean(dim=-1, keepdim=True)
    v = ((x - m) ** 2).mean(dim=-1, keepdim=True)
    return (x - m) / torch.sqrt(v + 1e-7)


class Wav2Vec2Model(nn.Module):
    def __init__(self, hidden=768, layers=12, ffn=3072, heads=12, conv_dim=512):
        super().__init__()
        self.masked_spec_embed = nn.Parameter(torch.empty(hidden).uniform_())
        self.feature_extractor = _FeatureExtractor(conv_dim)
        self.feature_projection = _FeatureProjection(conv_dim, hidden)
        self.encoder = _Encoder(hidden, layers, ffn, heads)
        self.heads = heads
        # the train-mode probabilities of facebook/wav2vec2-base-960h's configuration (what the reference's from_pretrained loads)
        self.config = types.SimpleNamespace(feat_proj_dropout=0.1, hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1,
                                            layerdrop=0.1, apply_spec_augment=True, mask_time_prob=0.05, mask_time_length=10)
        self.train_regularisers = True
        self.injected_masks = None   # an EncoderMasks to use instead of drawing (parity tests: masks recorded from the reference)
        self.static_rand = None      # a dropout.StepRandomness: the step's random state lives on the device (hipGraph-captured train steps)
        self.last_masks = None
        self.layers_run = 0          # Transformer layers executed / forwards since the counters were reset (LayerDrop skips layers: bench.py
        self.forwards_run = 0        # subtracts them from the step's flop count)

    compute_path = "hand-written HIP (a2f_mm bf16-MFMA GEMMs, MFMA flash attention, HIP LayerNorm / GELU / GroupNorm kernels)"

    @property
    def dtype_note(self):
        from .. import ops
        return ("encoder products as bf16x3 split MFMA (fp32-class), fp32 accumulate / softmax / LayerNorm" if ops.backend().mm_precision == 1
                else "encoder products as bf16 MFMA, fp32 accumulate / softmax / LayerNorm")

    def _step_masks(self, S, T, device):
        """The masks of this forward: None outside train mode; injected ones if set; else drawn from torch's CPU generator."""
        if not (self.training and self.train_regularisers):
            return None
        if self.injected_masks is not None:
            return self.injected_masks
        from .. import ops
        from ..dropout import draw_encoder_masks
        enc = self.encoder
        lay = enc.layers[0]
        if self.static_rand is not None:                       # captured step: nothing is drawn on the host here (trainer refreshes the device state)
            assert (self.static_rand.S, self.static_rand.T, self.static_rand.n_layers) == (S, T, len(enc.layers))
            return draw_encoder_masks(ops.backend(), S, T, len(enc.layers), self.heads, lay.layer_norm.normalized_shape[0],
                                      lay.feed_forward.intermediate_dense.out_features, self.config, None, device, static=self.static_rand)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return draw_encoder_masks(ops.backend(), S, T, len(enc.layers), self.heads, lay.layer_norm.normalized_shape[0],
                                  lay.feed_forward.intermediate_dense.out_features, self.config, seed, device)

    def _spec_augment(self, h, masks):
        """wav2vec.py:149-162: the frames SpecAugment picked are replaced by the learned masked_spec_embed (h: (S, T, d))."""
        if masks is None or masks.spec is None:
            return h
        return torch.where(masks.spec.view(h.shape[0], h.shape[1], 1), self.masked_spec_embed.to(h.dtype).view(1, 1, -1), h)

    def _forward_hip(self, wave, frame_num):
        """The encoder on the HIP path; wave (S, N) normalised -> (S, frame_num, hidden)."""
        from ..encoder_ops import ConvStackFn, DropoutFn, EncoderLayerFn, FeatEnc0Fn, LayerNormFn, LinearMMFn, PosConvFn
        S = wave.shape[0]
        fe = self.feature_extractor.conv_layers
        # feature encoder, in chunks of sequences: the sequences are independent there and an a2f_mm operand spans < 4 GiB (the layer-0
        # output of 32 sequences of 160 000 samples is 2.1 GB); the chunks are concatenated once the frame rate is down to 49 Hz
        n = wave.shape[1]
        from .. import ops as _ops
        L0, C0 = max((n - 10) // 5 + 1, 1), fe[0].conv.weight.shape[0]
        per = max(1, int(((1 << 31) - (1 << 20)) // (L0 * C0 * 4)))   # < 2 GiB per chunk: what the image-operand kernels' 32-bit epilogue offsets span
        per = -(-S // -(-S // per))                               # balanced chunks
        stack_w = [fe[i].conv.weight for i in range(1, len(fe))]
        outs = []
        for s0 in range(0, S, per):
            ns = min(per, S - s0)
            # the stack on split-32 image operands (encoder_ops.ConvStackFn): layer 0 then writes its output as the image layer 1 reads
            imgs = len(fe) > 1 and ConvStackFn.use_images(_ops.backend(), torch.empty(ns, L0, C0, device="meta"), stack_w)
            # layer 0 (1 -> 512 channels, k = 10, stride 5) + GroupNorm(512 groups) + GELU: 10 MACs per output, memory-bound
            z = FeatEnc0Fn.apply(wave[s0:s0 + per], fe[0].conv.weight, fe[0].layer_norm.weight, fe[0].layer_norm.bias, imgs)      # channels-last (s, L, 512)
            z = ConvStackFn.apply(z, tuple(CONV_STRIDE[1:len(fe)]), imgs, *stack_w)
            outs.append(z)
        z = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        if frame_num is not None:
            z = linear_interpolation(z, frame_num)                                # (S, T, 512)
        T, C = z.shape[1], z.shape[2]
        fp = self.feature_projection
        masks = self.last_masks = self._step_masks(S, T, z.device)
        h = LinearMMFn.apply(LayerNormFn.apply(z.reshape(S * T, C), fp.layer_norm.weight, fp.layer_norm.bias), fp.projection.weight, fp.projection.bias,
                             masks.featproj if masks is not None else None)
        enc = self.encoder
        pc = enc.pos_conv_embed
        d = h.shape[1]
        h = self._spec_augment(h.view(S, T, d), masks)
        h = PosConvFn.apply(h, pc.conv.weight, pc.conv.bias, pc.conv.groups).reshape(S * T, d)
        h = LayerNormFn.apply(h, enc.layer_norm.weight, enc.layer_norm.bias)
        if masks is not None and masks.enc is not None:
            h = DropoutFn.apply(h, masks.enc[0], masks.enc[1])
        self.forwards_run += 1
        h_img = None                                                         # the layer input as a split-32 image, handed from layer to layer (encoder_ops)
        for l, layer in enumerate(enc.layers):
            if masks is not None and masks.layer_skip[l]:
                continue                                                     # LayerDrop (HF Wav2Vec2Encoder.forward)
            self.layers_run += 1
            a, f = layer.attention, layer.feed_forward
            h_in, h_img_in = h, h_img
            h, h_img = EncoderLayerFn.apply(h, h_img, S, T, self.heads, masks.layers[l] if masks is not None else None,
                                            a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias, a.v_proj.weight,
                                            a.v_proj.bias, a.out_proj.weight, a.out_proj.bias, layer.layer_norm.weight, layer.layer_norm.bias,
                                            f.intermediate_dense.weight, f.intermediate_dense.bias, f.output_dense.weight, f.output_dense.bias,
                                            layer.final_layer_norm.weight, layer.final_layer_norm.bias)
            if masks is not None and masks.layer_keep_dev is not None:
                # LayerDrop decided on the device (captured step): the layer ran; a dropped one hands its input on and receives zero gradients
                keep_l = masks.layer_keep_dev[l]
                h = torch.where(keep_l, h, h_in)
                if h_img is not None:
                    h_img = torch.where(keep_l, h_img, h_img_in) if h_img_in is not None else None
        return h.view(S, T, d)

    def forward(self, input_values, dataset="vocaset", attention_mask=None, frame_num=None, **_):
        if dataset not in ("vocaset", None):
            raise NotImplementedError("only the vocaset branch (wav2vec.py:124-128) and the plain HF forward (dataset=None) are restated")
        if dataset is None:
            frame_num = None                                   # plain HF Wav2Vec2Model.forward: 49 Hz frames, no interpolation
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed by the reference FaceFormer (faceformer.py:149-151)")
        return types.SimpleNamespace(last_hidden_state=self._forward_hip(input_values, frame_num))
