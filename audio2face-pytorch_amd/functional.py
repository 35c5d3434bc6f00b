"""Autograd functions of the drop-in modules: each forward/backward is a fixed sequence of HIP launches.

Saved activations live in channels-last layout (frames, positions, channels) so every strided
convolution is an implicit GEMM over contiguous windows (see ops.conv_fwd).
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import ops
from .ops import ACT_NONE, ACT_TANH, AUX_ADD, AUX_RELU_BWD, AUX_TANH_BWD


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# =================================================================================================
# VocaLoss / FaceFormerLoss   (reference: src/loss/loss.py:4-55)
# =================================================================================================
class VocaLossFn(torch.autograd.Function):
    """(pred (B,n), gt (B,n)) -> out3 = [loss, rec_loss, vel_loss]; the backward gradient of out3 is consumed on device."""

    @staticmethod
    def forward(ctx, pred, gt, k_rec, k_vel):
        be = ops.backend()
        B, n = pred.shape
        p, g = _f32c(pred), _f32c(gt)
        out3 = torch.empty(3, device=p.device, dtype=torch.float32)
        be.vocaloss_fwd(p, g, B, n, float(k_rec), float(k_vel), out3)
        ctx.save_for_backward(p, g)
        ctx.k = (float(k_rec), float(k_vel))
        return out3

    @staticmethod
    def backward(ctx, g3):
        be = ops.backend()
        p, g = ctx.saved_tensors
        B, n = p.shape
        grad = torch.empty_like(p)
        be.vocaloss_bwd(p, g, _f32c(g3), B, n, ctx.k[0], ctx.k[1], grad)
        return grad, None, None, None


# =================================================================================================
# Vertex head: pred = H W^T + b + template   (decoder.4 / output_net.4 / vertice_map_r)
# =================================================================================================
class HeadFn(torch.autograd.Function):
    """H (B,K), W (n,K), b (n), template (B,n) or (1,n) broadcast over frames -> pred (B,n)."""

    @staticmethod
    def forward(ctx, H, W, b, template):
        be = ops.backend()
        B, K = H.shape
        n = W.shape[0]
        Hc, Wc, bc, tc = _f32c(H), _f32c(W), _f32c(b), _f32c(template).reshape(-1, n)
        pred = torch.empty(B, n, device=Hc.device, dtype=torch.float32)
        ops.linear_fwd(be, Hc, Wc, bc, pred, B, n, K, aux=tc, ldaux=(n if tc.shape[0] == B and B > 1 else 0), aux_mode=AUX_ADD,
                       tag="head_fwd")
        ctx.save_for_backward(Hc, Wc)
        ctx.trows = tc.shape[0]
        return pred

    @staticmethod
    def backward(ctx, G):
        be = ops.backend()
        Hc, Wc = ctx.saved_tensors
        B, K = Hc.shape
        n = Wc.shape[0]
        G = _f32c(G)
        dH = torch.zeros(B, K, device=G.device, dtype=torch.float32)
        be.gemm(G, Wc, dH, B, K, n, (0, n, 1), (0, K, 1), K, split_k=2, tag="head_dH")
        dW = torch.empty(n, K, device=G.device, dtype=torch.float32)
        db = torch.empty(n, device=G.device, dtype=torch.float32)
        ops.linear_bwd_weight(be, G, Hc, dW, db, B, n, K, tag="head_dW")
        dT = None
        if ctx.needs_input_grad[3]:
            dT = G if ctx.trows == B else G.sum(dim=0, keepdim=True)
        return dH, dW, db, dT


# =================================================================================================
# VOCA trunk   (reference: src/model/voca.py:38-48 up to the 50-wide bottleneck)
# =================================================================================================
VOCA_CONVS = ((37, 32, 8), (32, 32, 4), (32, 64, 2), (64, 64, 1))      # (Cin, Cout, positions out); input has 16 positions


def voca_is_features(x) -> bool:
    """True when ``x`` is the per-frame feature tensor (S, T, 29) rather than windows (B, 29, 16): the trunk then builds
    the 16-frame windows itself."""
    return x.dim() == 3 and x.shape[-1] == 29          # windows are (B, 29, 16): last dimension 16


def voca_frames(x) -> int:
    return x.shape[0] * x.shape[1] if voca_is_features(x) else x.shape[0]


def voca_trunk_fwd(be, x, one_hot, params):
    """x (B,29,16) windows - or (S,T,29) per-frame features, B = S*T, windows built on chip - one_hot (B,>=8),
    params = 8 conv tensors + 6 decoder tensors -> (h3 (B,50), saved tuple).

    One fused launch (csrc/voca_trunk.hip): the activations of a 32-frame block stay in LDS through all seven layers.
    ``layerwise=True`` in ``voca_trunk_fwd_layerwise`` keeps the per-layer implicit-GEMM sequence (used by tests to
    cross-check the fused kernel)."""
    feats = voca_is_features(x)
    if not hasattr(be, "voca_trunk_fwd"):
        if feats:
            from .data import feature_windows
            x = feature_windows(x)
        return voca_trunk_fwd_layerwise(be, x, one_hot, params)
    cw = [_f32c(p) for p in params[0:8]]
    fw0, fb0, fw1, fb1, fw3, fb3 = (_f32c(p) for p in params[8:14])
    B = voca_frames(x)
    if one_hot.shape[0] != B:
        raise ValueError(f"one_hot has {one_hot.shape[0]} rows for {B} frames")
    dev = x.device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    xc, oh = _f32c(x), _f32c(one_hot)
    a0, a1, a2, a3 = new(B, 16, 37), new(B, 8, 32), new(B, 4, 32), new(B, 2, 64)
    cat, z1, z2, h3 = new(B, 72), new(B, 72), new(B, 128), new(B, 50)
    if feats:
        be.voca_trunk_fwd_seq(xc, x.shape[0], x.shape[1], oh, oh.shape[1], cw[0::2], cw[1::2], [fw0, fw1, fw3], [fb0, fb1, fb3],
                              a0, a1, a2, a3, cat, z1, z2, h3)
    else:
        be.voca_trunk_fwd(xc, oh, B, oh.shape[1], cw[0::2], cw[1::2], [fw0, fw1, fw3], [fb0, fb1, fb3], a0, a1, a2, a3, cat, z1, z2, h3)
    return h3, (a0, a1, a2, a3, cat, z1, z2, fw0, fw1, fw3, cw[0], cw[2], cw[4], cw[6])


def voca_trunk_fwd_layerwise(be, x, one_hot, params):
    """Same function as voca_trunk_fwd, one implicit-GEMM launch per layer (reference sequence for the fused kernel)."""
    cw = params[0:8]                  # time_conv.{0,2,4,6}.{weight,bias}
    fw0, fb0, fw1, fb1, fw3, fb3 = (_f32c(p) for p in params[8:14])
    B = x.shape[0]
    dev = x.device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    xc, oh = _f32c(x), _f32c(one_hot)
    a0, cat = new(B, 16, 37), new(B, 72)
    be.voca_pack(xc, oh, B, oh.shape[1], a0, cat)
    acts = [a0]
    for li, (ci, co, pout) in enumerate(VOCA_CONVS):
        w, b = _f32c(cw[2 * li]), _f32c(cw[2 * li + 1])
        wg = new(co, 3 * ci)
        be.conv_weight_pack(w, wg, co, ci, 3, 0)
        last = pout == 1
        out = cat if last else new(B, pout, co)
        ops.conv_fwd(be, acts[-1], wg, b, out, B * pout, pout, ci, co, ldc=72 if last else co)
        acts.append(out)
    z1, z2, h3 = new(B, 72), new(B, 128), new(B, 50)
    ops.linear_fwd(be, cat, fw0, fb0, z1, B, 72, 72)                      # decoder.0
    ops.linear_fwd(be, z1, fw1, fb1, z2, B, 128, 72, act=ACT_TANH)        # decoder.1 + Tanh
    ops.linear_fwd(be, z2, fw3, fb3, h3, B, 50, 128)                      # decoder.3
    return h3, (a0, acts[1], acts[2], acts[3], cat, z1, z2, fw0, fw1, fw3, _f32c(cw[0]), _f32c(cw[2]), _f32c(cw[4]), _f32c(cw[6]))


# a2f_gemm_desc.precision of the VOCA trunk's batched weight gradients: 1 (default) = bf16x3 split products (the contraction
# runs over 1e4 .. 1e5 rows of smooth products: 1e-5-class error, 5.3x less matrix time), 0 = exact fp32 MFMA
VOCA_WGRAD_PRECISION = 1      # a2f_gemm_desc.precision of the trunk weight gradients: bf16x3 (0 = exact fp32 MFMA, 5.3x the matrix time)


def voca_trunk_bwd(be, saved, dh3, out=None):
    """Gradients of the 14 trunk parameters (same order as ``params``).  ``out``: optional list of 14 preallocated
    tensors (e.g. the trainer's flat-gradient views) that are OVERWRITTEN in place; otherwise new tensors are returned.

    One fused launch for the whole data-gradient chain (csrc/voca_trunk.hip) + one register-tile weight-gradient
    contraction per layer (csrc/wgrad.hip)."""
    if not hasattr(be, "voca_trunk_bwd_data"):
        return voca_trunk_bwd_layerwise(be, saved, dh3, out)
    a0, a1, a2, a3, cat, z1, z2, fw0, fw1, fw3, cw1, cw2, cw3, cw4 = saved
    B = cat.shape[0]
    dev = cat.device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    tgt = (lambda i, *s: out[i]) if out is not None else (lambda i, *s: new(*s))
    dh3 = _f32c(dh3)
    g_z2, g_z1, g_c4, g_c3, g_c2, g_c1 = new(B, 128), new(B, 72), new(B, 64), new(B, 2, 64), new(B, 4, 32), new(B, 8, 32)
    be.voca_trunk_bwd_data(dh3, z2, cat, a3, a2, a1, fw3, fw1, fw0, cw4, cw3, cw2, B, g_z2, g_z1, g_c4, g_c3, g_c2, g_c1)
    # all seven weight gradients (+ bias gradients through the ones column) in one launch pair; the heavy convolution
    # problems first so the light ones fill the tail
    grads = [None] * 14
    items = []
    old_prec = getattr(be, "wgrad_precision", 0)
    be.wgrad_precision = VOCA_WGRAD_PRECISION
    for li, (act_in, g) in enumerate(((a0, g_c1), (a1, g_c2), (a2, g_c3), (a3, g_c4))):
        ci, co, pout = VOCA_CONVS[li]
        dw, dbias = tgt(2 * li, co, ci, 3, 1), tgt(2 * li + 1, co)
        items.append(ops.conv_wgrad_item(be, act_in, g, dw, dbias, B * pout, pout, ci, co))
        grads[2 * li], grads[2 * li + 1] = dw, dbias
    dW0, db0, dW1, db1, dW3, db3 = tgt(8, 72, 72), tgt(9, 72), tgt(10, 128, 72), tgt(11, 128), tgt(12, 50, 128), tgt(13, 50)
    items.append(ops.linear_wgrad_item(be, g_z1, cat, dW0, db0, B, 72, 72))
    items.append(ops.linear_wgrad_item(be, g_z2, z1, dW1, db1, B, 128, 72))
    items.append(ops.linear_wgrad_item(be, dh3, z2, dW3, db3, B, 50, 128))
    be.wgrad_precision = old_prec
    be.wgrad_batch(items)
    grads[8:14] = [dW0, db0, dW1, db1, dW3, db3]
    return grads


def voca_trunk_bwd_layerwise(be, saved, dh3, out=None):
    """Same gradients, one implicit-GEMM launch per layer and direction (reference sequence for the fused kernel)."""
    a0, a1, a2, a3, cat, z1, z2, fw0, fw1, fw3, cw1, cw2, cw3, cw4 = saved
    B = cat.shape[0]
    dev = cat.device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    wgs = []                                              # packed (Co, 3*Ci) conv weights for the data gradients
    for li, w in enumerate((cw1, cw2, cw3, cw4)):
        ci, co, _ = VOCA_CONVS[li]
        wg = new(co, 3 * ci)
        if li > 0:                                        # conv 1 needs no data gradient
            be.conv_weight_pack(w, wg, co, ci, 3, 0)
        wgs.append(wg)
    tgt = (lambda i, *s: out[i]) if out is not None else (lambda i, *s: new(*s))
    dh3 = _f32c(dh3)
    grads = [None] * 14
    # decoder.3 (128 -> 50)
    dW3, db3 = tgt(12, 50, 128), tgt(13, 50)
    ops.linear_bwd_weight(be, dh3, z2, dW3, db3, B, 50, 128)
    dz2 = new(B, 128)
    ops.linear_bwd_input(be, dh3, fw3, dz2, B, 50, 128, aux=z2, ldaux=128, aux_mode=AUX_TANH_BWD)
    # decoder.1 (72 -> 128)
    dW1, db1 = tgt(10, 128, 72), tgt(11, 128)
    ops.linear_bwd_weight(be, dz2, z1, dW1, db1, B, 128, 72)
    dz1 = new(B, 72)
    ops.linear_bwd_input(be, dz2, fw1, dz1, B, 128, 72)
    # decoder.0 (72 -> 72); only the 64 conv features need an input gradient (ReLU of conv 4 applied here)
    dW0, db0 = tgt(8, 72, 72), tgt(9, 72)
    ops.linear_bwd_weight(be, dz1, cat, dW0, db0, B, 72, 72)
    d_act = new(B, 64)
    ops.linear_bwd_input(be, dz1, fw0, d_act, B, 72, 72, n_cols=64, aux=cat, ldaux=72, aux_mode=AUX_RELU_BWD)
    grads[8:14] = [dW0, db0, dW1, db1, dW3, db3]
    # temporal conv stack, last to first
    acts = (a0, a1, a2, a3)
    for li in (3, 2, 1, 0):
        ci, co, pout = VOCA_CONVS[li]
        R = B * pout
        dwg, dbias = new(co, 3 * ci), tgt(2 * li + 1, co)
        ops.conv_bwd_weight(be, acts[li], d_act, dwg, dbias, R, pout, ci, co)
        dw = tgt(2 * li, co, ci, 3, 1)
        be.conv_weight_pack(dwg, dw, co, ci, 3, 1)
        grads[2 * li], grads[2 * li + 1] = dw, dbias
        if li > 0:
            d_in = new(B, 2 * pout, ci)
            ops.conv_k3s2_bwd_input(be, d_act, wgs[li], d_in, R, pout, ci, co, relu_of=acts[li])
            d_act = d_in
    return grads


class VocaTrunkFn(torch.autograd.Function):
    """x (B,29,16), one_hot (B,>=8), 8 conv tensors, 6 decoder tensors -> h3 (B,50)."""

    @staticmethod
    def forward(ctx, x, one_hot, *params):
        h3, saved = voca_trunk_fwd(ops.backend(), x, one_hot, params)
        ctx.save_for_backward(*saved)
        return h3

    @staticmethod
    def backward(ctx, dh3):
        return (None, None, *voca_trunk_bwd(ops.backend(), ctx.saved_tensors, dh3))


# =================================================================================================
# Audio2Mesh trunk   (reference: src/model/audio2face.py:57-65 up to the 50-wide bottleneck)
# =================================================================================================
A2M_CH = (1, 72, 108, 162, 243, 256)           # analysis channels; width 32 -> 16 -> 8 -> 4 -> 2 -> 1   (audio2face.py:13-29)
A2M_BN_MOMENTUM, A2M_BN_EPS = 0.1, 1e-5


# =================================================================================================
# Audio2Mesh trunk on the bf16-MFMA GEMM family (a2f_mm): convolutions 1..9 as windowed implicit
# GEMMs with bf16x3 split products.  Channel counts that are not multiples of 4 (162, 243) are stored with a padded channel
# stride (164, 244; the pad channels are identically zero through conv / BatchNorm / ReLU and carry zero gradients), so every
# window starts on a 16-byte boundary.
# =================================================================================================
A2M_CHP = (1, 72, 108, 164, 244, 256)          # padded analysis channels


# Padded / re-laid-out copies of parameters are refreshed every step (the parameters change).  Their BUFFERS belong to the model (`A2MPackBuffers`, kept by
# Audio2Mesh in its __dict__ per device: they live exactly as long as the module, which is what a captured hipGraph with their addresses baked in and an
# autograd graph that saved them both hold), and ONE a2f_relayout_batch launch fills all of them - nine packed convolution weights, their transposes for the
# data-gradient products, two padded biases, eight padded BatchNorm vectors - where rounds 3-5 issued ~40 copies / fills of ~5 us each (a cache keyed on the
# tensor objects never hit inside an autograd Function: its inputs are fresh aliases on every call).  The nine weight gradients take the way back in one launch too.
A2M_PAD_LAYERS = (2, 3)                          # analysis blocks whose output channel count is padded (162 -> 164, 243 -> 244)


class A2MPackBuffers:
    def __init__(self, device, KH9: int):
        new = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
        self.device, self.KH9 = device, KH9
        self.dims = {}                                                         # layer -> (Co, Ci, KH, Cop, Cip)
        for l in range(1, 5):
            self.dims[l] = (A2M_CH[l + 1], A2M_CH[l], 3, A2M_CHP[l + 1], A2M_CHP[l])
        for l in range(5, 9):
            self.dims[l] = (256, 256, 3, 256, 256)
        self.dims[9] = (256, 256, KH9, 256, 256)
        self.wg = {l: new(d[3], d[2] * d[4]) for l, d in self.dims.items()}       # (Cop, KH * Cip), k = kh * Cip + ci
        self.wgT = {l: new(d[2] * d[4], d[3]) for l, d in self.dims.items()}      # its transpose: both operands of a data-gradient product k-contiguous
        self.bias = {l: new(A2M_CHP[l + 1]) for l in A2M_PAD_LAYERS}
        self.bn = {l: tuple(new(A2M_CHP[l + 1]) for _ in range(4)) for l in A2M_PAD_LAYERS}     # gamma (pad 1), beta (0), running_mean (0), running_var (1)

    def refresh_items(self, convs, bns, bufs):
        items = []
        for l, (Co, Ci, KH, Cop, Cip) in self.dims.items():
            w = convs[l][0]
            items.append((w, self.wg[l], Co, Ci, KH, Cop, Cip, 0))
            items.append((w, self.wgT[l], Co, Ci, KH, Cop, Cip, 1))
        for l in A2M_PAD_LAYERS:
            C, Cp = A2M_CH[l + 1], A2M_CHP[l + 1]
            items.append((convs[l][1], self.bias[l], C, 1, 1, Cp, 1, 0))
            for src, dst, pad in zip((bns[l][0], bns[l][1], bufs[2 * l], bufs[2 * l + 1]), self.bn[l], (1.0, 0.0, 0.0, 1.0)):
                items.append((src, dst, C, 1, 1, Cp, 1, 0, pad))
        return items


# a2f_mm_desc.precision of the trunk's products on a2f_mm, per kind: "fwd,wgrad,dgrad" (or one number for all three); 2 = bf16x6
# (fp32-class), 1 = bf16x3.  Default 2,1,1: the FORWARD convolutions need the fp32-class form - their rounding reaches the batch
# statistics of the next BatchNorm, whose (z - mean) / std amplifies it: with a bf16x3 forward the weight gradients of the analysis stack
# sit 1.3e-2 from the reference on the B = 4 fixture (G2) whatever the backward products use - while the weight- and data-gradient
# products are linear in their operands and hold 6e-5 / 2e-5 in bf16x3 (round 3 A-B on the GPU, tools/a2m_grad_errors.py: all-bf16x6
# 2.1e-5 worst at 8.2 ms per B = 1024 step; 2,1,1: 6.0e-5 at 6.7 ms; 1,x,x: 1.3e-2).
_a2m_prec = [int(v) for v in os.environ.get("A2F_A2M_MM_PRECISION", "2,1,1").split(",")]
A2M_PREC_FWD, A2M_PREC_WGRAD, A2M_PREC_DGRAD = (_a2m_prec * 3)[:3] if len(_a2m_prec) == 1 else tuple(_a2m_prec)
A2M_PREC_FWD_LAYER = {}        # layer (1..9) -> precision of that convolution's forward product where it differs from A2M_PREC_FWD (tools/a2m_layer_precision_probe.py)



@contextlib.contextmanager
def _mm_precision(be, prec):
    old = be.mm_precision
    be.mm_precision = prec
    try:
        yield
    finally:
        be.mm_precision = old


class A2MTrunkMMFn(torch.autograd.Function):
    """x (B,52,32), one_hot (B,n_onehot), training flag, 20 BN buffers (10 x running_mean, running_var in layer order), then the parameters:
    10 conv (w, b) pairs in layer order, 10 BN (gamma, beta) pairs, 3 Linear (w, b) pairs -> h3 (B,50).  Activations are channels-last:
    analysis stage (B*64 rows-of-W, W, C), articulation stage (B, H, 256)."""

    @staticmethod
    def forward(ctx, x, one_hot, training, pk, *rest):
        be = ops.backend()
        with _mm_precision(be, A2M_PREC_FWD):
            return A2MTrunkMMFn._forward(ctx, be, x, one_hot, training, pk, *rest)

    @staticmethod
    def _forward(ctx, be, x, one_hot, training, pk, *rest):
        bufs, params = rest[:20], rest[20:]
        convs = [(_f32c(params[2 * i]), _f32c(params[2 * i + 1])) for i in range(10)]
        bns = [(_f32c(params[20 + 2 * i]), _f32c(params[21 + 2 * i])) for i in range(10)]
        fcs = [(_f32c(params[40 + 2 * i]), _f32c(params[41 + 2 * i])) for i in range(3)]
        be.relayout_batch(pk.refresh_items(convs, bns, bufs))          # every packed / padded copy of this step's parameters: one launch
        B, no = x.shape[0], one_hot.shape[1]
        H0 = 52 + no
        dev = x.device
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        ncat = 256 + no
        a0, cat = new(B, H0, 32), new(B, ncat)
        be.a2m_pack(_f32c(x), _f32c(one_hot), B, no, a0, cat, ncat, 256)
        sv = {"wg": [None] * 10, "conv_in": [None] * 10, "bn_x": [None] * 10, "bn_b": [None] * 10, "stats": [None] * 10, "bn_g": [None] * 10}

        def bn(i, xin, R, C, Cp, relu):
            gp, bp, rmp, rvp = pk.bn[i] if Cp != C else (bns[i][0], bns[i][1], bufs[2 * i], bufs[2 * i + 1])
            y, mu, istd = new(R, Cp), new(Cp), new(Cp)
            be.bn_fwd(xin, R, Cp, gp, bp, rmp, rvp, A2M_BN_MOMENTUM, A2M_BN_EPS, training, relu, y, mu, istd)
            sv["bn_x"][i], sv["bn_b"][i], sv["stats"][i], sv["bn_g"][i] = xin, bp, (mu, istd), gp
            return y

        # analysis layer 0 (one input channel, K = 3) and its BatchNorm: the convolution is three exact-fp32 multiply-adds per output from an input 72x
        # smaller than its output, so it is RECOMPUTED inside every BatchNorm pass instead of stored (a2f_bn_fwd_conv0 / a2f_bn_bwd_conv0; with Ci = 1 the
        # packed weight (Co, KH * Ci) is the parameter's own memory).  Same box: 5.82 -> 5.73 ms per step against storing the layer's output with a streaming kernel
        # of its own (which had replaced the K = 3 implicit GEMM: 6.00 -> 5.76).
        w0, b0 = convs[0]
        wg0 = w0.view(72, 3)
        R = B * H0 * 16
        sv["wg"][0], sv["conv_in"][0] = wg0, a0
        g0, bt0 = bns[0]
        act, mu0, istd0 = new(R, 72), new(72), new(72)
        be.bn_fwd_conv0(a0, wg0, b0, 16, R, 72, g0, bt0, bufs[0], bufs[1], A2M_BN_MOMENTUM, A2M_BN_EPS, training, True, act, mu0, istd0)
        sv["bn_x"][0], sv["bn_b"][0], sv["stats"][0], sv["bn_g"][0] = (a0, wg0, b0), bt0, (mu0, istd0), g0
        Wd = 16
        for l in range(1, 5):                                  # analysis layers 1..4: windows over W, sequences = (frame, row)
            ci, co, cip, cop = A2M_CH[l], A2M_CH[l + 1], A2M_CHP[l], A2M_CHP[l + 1]
            wg = pk.wg[l]
            Wo = Wd // 2
            R = B * H0 * Wo
            z = new(R, cop)
            ops.mm_conv_fwd(be, act, wg, z, B * H0, Wd, Wo, cip, cop, 3, 2, pad=1, bias=pk.bias[l] if cop != co else convs[l][1], precision=A2M_PREC_FWD_LAYER.get(l))
            sv["wg"][l], sv["conv_in"][l] = wg, act
            act = bn(l, z, R, co, cop, True)
            Wd = Wo
        Hd = H0
        for l in range(5, 8):                                  # articulation: windows over H, sequences = frames
            wg, b = pk.wg[l], convs[l][1]
            Ho = Hd // 2
            z = new(B * Ho, 256)
            ops.mm_conv_fwd(be, act, wg, z, B, Hd, Ho, 256, 256, 3, 2, pad=1, bias=b, precision=A2M_PREC_FWD_LAYER.get(l))
            sv["wg"][l], sv["conv_in"][l] = wg, act
            act = bn(l, z, B * Ho, 256, 256, True)
            Hd = Ho
        act = bn(8, act, B * Hd, 256, 256, False)              # BN -> conv -> ReLU   (order flip, audio2face.py:41-43)
        wg, b = pk.wg[8], convs[8][1]
        Ho = Hd // 2
        z8 = new(B * Ho, 256)
        ops.mm_conv_fwd(be, act, wg, z8, B, Hd, Ho, 256, 256, 3, 2, pad=1, bias=b, act=ops.MM_ACT_RELU, precision=A2M_PREC_FWD_LAYER.get(8))
        sv["wg"][8], sv["conv_in"][8] = wg, act
        Hd = Ho
        zb = bn(9, z8, B * Hd, 256, 256, False)                # BN -> conv(4,1)/s4 -> ReLU  (audio2face.py:44-46)
        wg, b, KH9 = pk.wg[9], convs[9][1], pk.KH9
        be.mm(ops.MMOp(zb, KH9 * 256), ops.MMOp(wg, KH9 * 256), cat, B, 256, KH9 * 256, ncat, bias=b, act=ops.MM_ACT_RELU)
        sv["wg"][9], sv["conv_in"][9] = wg, zb
        z1, z2, h3 = new(B, 72), new(B, 128), new(B, 50)
        # output_net.0 / .1 / .3 on a2f_mm as well (1024 x 72 x 268 takes 80 us on the exact-fp32 generic GEMM, a few us here); only the BACKWARD of the
        # 50-wide layer keeps the generic GEMM: there the 50-float rows of dh3 are the k-contiguous operand and they are not 16-byte aligned
        ops.mm_linear_fwd(be, cat, fcs[0][0], fcs[0][1], z1, B, 72, ncat)
        ops.mm_linear_fwd(be, z1, fcs[1][0], fcs[1][1], z2, B, 128, 72)
        torch.tanh_(z2)
        ops.mm_linear_fwd(be, z2, fcs[2][0], fcs[2][1], h3, B, 50, 128)
        if training:                                         # the padded layers' running statistics go back into the module's buffers: one launch for the four
            be.relayout_batch([(src, bufs[2 * l + j], A2M_CH[l + 1], 1, 1, A2M_CHP[l + 1], 1, 2) for l in A2M_PAD_LAYERS for j, src in enumerate(pk.bn[l][2:])])
        ctx.sv, ctx.pk = sv, pk
        ctx.misc = (B, no, H0, Hd, KH9, training, cat, z8, z1, z2, [f[0] for f in fcs])
        return h3

    @staticmethod
    def backward(ctx, dh3):
        be = ops.backend()
        # eval-mode BatchNorm (running statistics: no per-batch renormalisation of the gradient scale) is only differentiated by parity tests:
        # there every product keeps the fp32-class form (bf16x3 data gradients sat at 1.7e-4 on the eval-mode fixture, bf16x6 at 2.5e-5)
        training = ctx.misc[5]
        ctx.prec_wgrad, prec_dgrad = (A2M_PREC_WGRAD, A2M_PREC_DGRAD) if training else (2, 2)
        with _mm_precision(be, prec_dgrad):
            return A2MTrunkMMFn._backward(ctx, be, dh3)

    @staticmethod
    def _backward(ctx, be, dh3):
        sv, pk = ctx.sv, ctx.pk
        B, no, H0, Hlast, KH9, training, cat, z8, z1, z2, fw = ctx.misc
        unpack = []                                           # (packed weight gradient, gradient in the parameter's layout, layer): one a2f_relayout_batch at the end
        dev = cat.device
        ncat = 256 + no
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        zeros = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        dh3 = _f32c(dh3)
        g_conv, g_bn, g_fc = [None] * 20, [None] * 20, [None] * 6
        dW, db = new(50, 128), new(50)
        ops.linear_bwd_weight(be, dh3, z2, dW, db, B, 50, 128); g_fc[4], g_fc[5] = dW, db
        dz2 = new(B, 128)
        ops.linear_bwd_input(be, dh3, fw[2], dz2, B, 50, 128, aux=z2, ldaux=128, aux_mode=AUX_TANH_BWD)
        dW, db = new(128, 72), new(128)
        ops.mm_linear_wgrad(be, dz2, z1, dW, B, 128, 72, db=db); g_fc[2], g_fc[3] = dW, db
        dz1 = new(B, 72)
        be.mm(ops.MMOp(dz2, 128), ops.MMOp(fw[1], 72, trans=True), dz1, B, 72, 128, 72)
        dW, db = new(72, ncat), new(72)
        ops.mm_linear_wgrad(be, dz1, cat, dW, B, 72, ncat, db=db); g_fc[0], g_fc[1] = dW, db
        d = new(B, 256)                                       # grad w.r.t. conv-9 pre-activation (ReLU mask of cat fused): the first 256 input columns of output_net.0
        be.mm(ops.MMOp(dz1, 72), ops.MMOp(fw[0], ncat, trans=True), d, B, 256, 72, 256, aux=cat, ldaux=ncat, aux_mode=ops.MM_AUX_RELU_BWD)


        def bn_bwd(i, dy, R, C, Cp, relu):
            mu, istd = sv["stats"][i]
            dx, dg, dbt = new(R, Cp), new(Cp), new(Cp)
            be.bn_bwd(sv["bn_x"][i], sv["bn_b"][i], dy, R, Cp, sv["bn_g"][i], mu, istd, training, relu, dx, dg, dbt)
            g_bn[2 * i], g_bn[2 * i + 1] = dg[:C], dbt[:C]
            return dx

        # conv 9: (KH9,1)/s KH9, no pad: a plain GEMM over the (B, KH9*256) rows of zb
        dwg, dbv = new(256, KH9 * 256), new(256)
        with _mm_precision(be, ctx.prec_wgrad):
            ops.mm_linear_wgrad(be, d, sv["conv_in"][9], dwg, B, 256, KH9 * 256, db=dbv)  # bias gradients: row sums inside the weight-gradient product
        g_conv[18], g_conv[19] = new(256, 256, KH9, 1), dbv
        unpack.append((dwg, g_conv[18], 9))
        dzb = new(B, KH9 * 256)
        ops.mm_linear_dgrad(be, d, sv["wg"][9], dzb, B, 256, KH9 * 256, WT=pk.wgT[9])
        dz = bn_bwd(9, dzb, B * Hlast, 256, 256, False)
        dz = torch.ops.aten.threshold_backward(dz, z8, 0.0)   # conv 8's ReLU (z8 is its output): one launch
        Hd = 2 * Hlast

        def conv_bwd(i, dout, S, Lin, Lout, ci, co, cip, cop, shape, need_dx=True):
            dwg, dbv = new(cop, 3 * cip), new(cop)
            with _mm_precision(be, ctx.prec_wgrad):
                ops.mm_conv_wgrad(be, sv["conv_in"][i], dout, dwg, S, Lin, Lout, cip, cop, 3, 2, pad=1, db=dbv)
            g_conv[2 * i], g_conv[2 * i + 1] = new(*shape), dbv[:co]
            unpack.append((dwg, g_conv[2 * i], i))
            if not need_dx:
                return None
            din = new(S * Lin, cip)                           # every position is written (Lin = 2 Lout, pad = 1)
            ops.mm_conv_dgrad(be, dout, sv["wg"][i], din, S, Lin, Lout, cip, cop, 3, 2, pad=1, WgT=pk.wgT[i])
            return din

        dact = conv_bwd(8, dz, B, Hd, Hlast, 256, 256, 256, 256, (256, 256, 3, 1))
        d = bn_bwd(8, dact, B * Hd, 256, 256, False)
        for l in (7, 6, 5):                                   # articulation conv -> BN -> ReLU blocks
            draw = bn_bwd(l, d, B * Hd, 256, 256, True)
            d = conv_bwd(l, draw, B, 2 * Hd, Hd, 256, 256, 256, 256, (256, 256, 3, 1))
            Hd *= 2
        Wd = 1
        for l in (4, 3, 2, 1):                                # analysis layers on the MFMA GEMM
            ci, co, cip, cop = A2M_CH[l], A2M_CH[l + 1], A2M_CHP[l], A2M_CHP[l + 1]
            draw = bn_bwd(l, d, B * H0 * Wd, co, cop, True)
            d = conv_bwd(l, draw, B * H0, 2 * Wd, Wd, ci, co, cip, cop, (co, ci, 1, 3))
            Wd *= 2
        # analysis layer 0 (Ci = 1): exact-fp32 generic kernels; no data gradient needed
        a0_, wg0_, cb0_ = sv["bn_x"][0]                      # layer 0: the convolution output is recomputed, not read
        mu0, istd0 = sv["stats"][0]
        # ... and neither is its pre-activation gradient: the layer has no data gradient, so the same pass that forms it collects the layer's weight / bias gradient
        # (a2f_bn_bwd_conv0 with dwg / dcbias; with Ci = 1 the (C, 3) result IS the parameter's (72, 1, 1, 3) layout) - 302 MB not written and not read back at B = 1024
        dg0, dbt0, dw0, dbias0 = new(72), new(72), new(72, 1, 1, 3), new(72)
        be.bn_bwd_conv0(a0_, wg0_, cb0_, Wd, sv["bn_b"][0], d, B * H0 * Wd, 72, sv["bn_g"][0], mu0, istd0, training, True, None, dg0, dbt0, dwg=dw0, dcbias=dbias0)
        g_bn[0], g_bn[1] = dg0, dbt0
        g_conv[0], g_conv[1] = dw0, dbias0
        be.relayout_batch([(dwg, g, *pk.dims[l], 2) for dwg, g, l in unpack])
        ctx.sv = ctx.pk = None
        return (None, None, None, None) + (None,) * 20 + tuple(g_conv) + tuple(g_bn) + tuple(g_fc)


# (Round 1's trunk on the generic split GEMM of gemm.hip - 11.4 ms at B = 1024 against 7.9 ms - was removed in round 3; a2f_mm with
# bf16x6 products is the only Audio2Mesh trunk.  A2F_A2M_MM_PRECISION=1 (bf16x3, 7.6 ms) gives the same predictions (9e-6) but the
# batch-statistics BatchNorm backward amplifies its 2^-16 product error into 1e-2 on the deepest weight gradients of the B = 4 fixture -
# outside the parity bar, so it is opt-in.)


# =================================================================================================
# Fused head + loss + head backward (single pass over the frame matrix; see csrc/head_fused.hip)
# =================================================================================================
class HeadLossFusedFn(torch.autograd.Function):
    """(H (B,K), W (n,K), b (n), template (B,n)|(1,n), gt (B,n), Bloss, k_rec, k_vel) -> (pred (B,n), out3).

    out3 = [loss, rec_loss, vel_loss].  The head gradients for d(loss) are produced by the same kernel pass that
    writes pred (G = dloss/dpred never exists in memory); backward only scales them by the incoming d/d(loss).
    pred is returned for metrics / prediction and is NOT differentiable on this path, nor are rec/vel separately
    (use HeadFn + VocaLossFn for arbitrary graphs).
    """

    @staticmethod
    def forward(ctx, H, W, b, template, gt, Bloss, k_rec, k_vel, unit_grad=False):
        be = ops.backend()
        ctx.set_materialize_grads(False)          # pred is not differentiable: never build a (B,n) zero gradient for it
        ctx.unit_grad = bool(unit_grad)           # caller promises d/d(loss) == 1 (loss["loss"].backward()): skip the rescale
        B, K = H.shape
        n = W.shape[0]
        Hc, Wc, bc = _f32c(H), _f32c(W), _f32c(b)
        tc, gc = _f32c(template).reshape(-1, n), _f32c(gt).reshape(B, n)
        dev = Hc.device
        pred = torch.empty(B, n, device=dev, dtype=torch.float32)
        out3 = torch.empty(3, device=dev, dtype=torch.float32)
        dH = torch.empty(B, K, device=dev, dtype=torch.float32)
        dW = torch.empty(n, K, device=dev, dtype=torch.float32)
        db = torch.empty(n, device=dev, dtype=torch.float32)
        be.head_fused(Hc, Wc, bc, tc, tc.shape[0], gc, B, int(Bloss), n, K, float(k_rec), float(k_vel), pred, out3, dH, dW, db)
        ctx.save_for_backward(dH, dW, db)
        ctx.mark_non_differentiable(pred)
        return pred, out3

    @staticmethod
    def backward(ctx, _gpred, g3):
        dH, dW, db = ctx.saved_tensors
        if g3 is None:
            return (None,) * 9
        if ctx.unit_grad:
            return dH, dW, db, None, None, None, None, None, None
        g = g3[0]                        # d/d(loss); stays on device
        return dH * g, dW * g, db * g, None, None, None, None, None, None


# =================================================================================================
# Generic Linear / matmul on the HIP GEMM (FaceFormer projections, collapsed feedback matrix)
# =================================================================================================
class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b);  x (M,K), W (N,K), b (N) or None."""

    @staticmethod
    def forward(ctx, x, W, b):
        be = ops.backend()
        xc, Wc = _f32c(x), _f32c(W)
        bc = _f32c(b) if b is not None else None
        M, K = xc.shape
        N = Wc.shape[0]
        y = torch.empty(M, N, device=xc.device, dtype=torch.float32)
        ops.linear_fwd(be, xc, Wc, bc, y, M, N, K)
        ctx.save_for_backward(xc, Wc)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        be = ops.backend()
        xc, Wc = ctx.saved_tensors
        M, K = xc.shape
        N = Wc.shape[0]
        dy = _f32c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            ops.linear_bwd_input(be, dy, Wc, dx, M, N, K)
        dW = torch.empty(N, K, device=dy.device, dtype=torch.float32)
        db = torch.empty(N, device=dy.device, dtype=torch.float32)
        ops.linear_bwd_weight(be, dy, xc, dW, db, M, N, K)
        return dx, dW, (db if ctx.has_bias else None)


class MatmulFn(torch.autograd.Function):
    """C = A B;  A (M,K), B (K,N), both row-major."""

    @staticmethod
    def forward(ctx, A, B):
        be = ops.backend()
        Ac, Bc = _f32c(A), _f32c(B)
        M, K = Ac.shape
        N = Bc.shape[1]
        Cm = torch.empty(M, N, device=Ac.device, dtype=torch.float32)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        sk = ops.split_k_for(K, tiles)
        be.gemm(Ac, Bc, Cm, M, N, K, (0, K, 1), (0, N, 1), N, split_k=sk, splitk_ws=ops._splitk_ws(Cm, sk, M, N))
        ctx.save_for_backward(Ac, Bc)
        return Cm

    @staticmethod
    def backward(ctx, dC):
        be = ops.backend()
        Ac, Bc = ctx.saved_tensors
        M, K = Ac.shape
        N = Bc.shape[1]
        dC = _f32c(dC)
        dA = torch.empty(M, K, device=dC.device, dtype=torch.float32)
        be.gemm(dC, Bc, dA, M, K, N, (0, N, 1), (0, 1, N), K)                    # dA = dC B^T
        dB = torch.empty(K, N, device=dC.device, dtype=torch.float32)
        be.gemm(Ac, dC, dB, K, N, M, (0, 1, K), (0, N, 1), N)                    # dB = A^T dC
        return dA, dB


# =================================================================================================
# FaceFormer decoder recurrence   (reference: src/model/faceformer.py:154-185, evaluated incrementally)
# =================================================================================================
class FFDecoderFn(torch.autograd.Function):
    """(params block (A2F_FF_PARAM_FLOATS,), pe (period,64), cross (S,T,64), style (S,64)) -> y (S,T,64)."""

    @staticmethod
    def forward(ctx, params, pe, cross, style, period, drop=None):
        """drop = (row_bits, attn_bits, keep_scale): train-mode variant with one dropout mask per frame row (a2f_ff_decoder_fwd_drop)."""
        be = ops.backend()
        pc, pec, cc, sc = _f32c(params), _f32c(pe), _f32c(cross), _f32c(style)
        S, T, _ = cc.shape
        y = torch.empty(S, T, 64, device=cc.device, dtype=torch.float32)
        save = torch.empty(be.ff_save_floats(S, T), device=cc.device, dtype=torch.float32)
        if drop is None:
            be.ff_decoder_fwd(pc, pec, cc, sc, S, T, int(period), y, save)
        else:
            be.ff_decoder_fwd(pc, pec, cc, sc, S, T, int(period), y, save, drop=drop)
        ctx.save_for_backward(pc, save)
        ctx.dims = (S, T, int(period))
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        be = ops.backend()
        pc, save = ctx.saved_tensors
        S, T, period = ctx.dims
        dy = _f32c(dy)
        dparams = torch.empty_like(pc)            # overwritten (per-sequence slabs folded in fixed order)
        dcross = torch.empty(S, T, 64, device=dy.device, dtype=torch.float32)
        dstyle = torch.empty(S, 64, device=dy.device, dtype=torch.float32)
        if ctx.drop is None:
            be.ff_decoder_bwd(pc, dy, S, T, period, save, dparams, dcross, dstyle)
        else:
            be.ff_decoder_bwd(pc, dy, S, T, period, save, dparams, dcross, dstyle, drop=ctx.drop)
        return dparams, None, dcross, dstyle, None, None


# =================================================================================================
# Linear interpolation along time   (reference: src/model/wav2vec.py:76-84 linear_interpolation)
# =================================================================================================
class LinearInterpFn(torch.autograd.Function):
    """features (N, L_in, C) -> (N, L_out, C), align_corners=True linear interpolation along L (csrc/interp.hip):
    the same function as ``F.interpolate(x.transpose(1, 2), size=L_out, align_corners=True, mode="linear").transpose(1, 2)``."""

    @staticmethod
    def forward(ctx, x, l_out: int):
        x = _f32c(x)
        N, L, C_ = x.shape
        y = torch.empty(N, l_out, C_, device=x.device, dtype=torch.float32)
        ops.backend().interp_linear(x, N, L, C_, l_out, y)
        ctx.dims = (N, L, C_, l_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, L, C_, l_out = ctx.dims
        dy = _f32c(dy)
        dx = torch.empty(N, L, C_, device=dy.device, dtype=torch.float32)
        ops.backend().interp_linear(dy, N, L, C_, l_out, dx, backward=True)
        return dx, None
