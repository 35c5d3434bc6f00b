"""Autograd functions of the wav2vec2 encoder on the HIP path (HF Wav2Vec2Model as wrapped by
/root/reference/src/model/wav2vec.py:91-189): every dense contraction runs on ``a2f_mm`` (bf16-MFMA GEMM family), attention
on ``a2f_attn_*`` (MFMA flash attention), LayerNorm / bias gradients / GELU' on the row kernels - forward and hand-written
backward, fixed launch sequences, deterministic reductions.

Activations are channels-last rows: (sequence, frame, channel) flattened to (rows, channels).  Precision follows
``ops.backend().mm_precision`` (1 = bf16x3 split products, fp32-class - the parity setting; 0 = bf16 operands for
``percision: 16-mixed``); storage, softmax, LayerNorm statistics and every accumulation are fp32.
"""
from __future__ import annotations

import torch

from . import ops
from .ops import MM_ACT_GELU, MM_ACT_GELU_SAVE_GRAD, MM_ACT_NONE, MM_AUX_ADD, MM_AUX_GELU_BWD, MM_AUX_MUL_LIKE_C, MMOp

LN_EPS = 1e-5
import os as _os

# Module switches (plain attributes: the tests set them; none is an environment knob).  Rounds 2-4 carried an A-B variant behind most forms below (pre-activation
# saved + streaming GELU' pass, three q / k / v launches, the positional convolution as batched a2f_mm launches, attention on fp32 q|k|v beside image GEMMs ...):
# measured slower every time they were compared (docs/DESIGN_rounds_1-3.md) and deleted in round 5.
MM_IMG = True                    # the Transformer layers' and the feature-encoder stack's products on split-32 image operands (from IMG_MIN_ROWS rows)
WIMG_CACHE = True                # a layer's weight images are built once per state of the weights
WIMG_VERIFY = _os.environ.get("A2F_WIMG_VERIFY") == "1"           # debug guard: every cache hit re-splits the weights and compares (raises on stale images)
WEIGHTS_EPOCH = 0                # bumped by whoever rewrites parameters behind torch's back (trainer.FlatAdam.step)


def bump_weights_epoch():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


IMG_MIN_ROWS = 2048            # below this many rows the 256 x 256 image tiles leave the chip empty: fp32-operand kernels
# Weight gradients on a second stream beside the data-gradient chain.  History: paid 1 ms of 157 in round 2 (tile tails of the 128 x 128 kernels filled each
# other); with the 256 x 256 image kernels on the unsplit batch it measured 131.0 (on) vs 129.9 ms (off) and, while every TrainStep of a process still made its
# own capture stream, aliased a hardware queue inside the full bench run (+25 ms): off.  Since the step runs as two half-batches (19 200 rows: the N = 768
# launches have 225 tiles for 256 CUs) the lane fills those idle CUs: 122.6 -> 119.8 ms standalone, 119.2 -> 117.4 ms inside the full bench run, 12 x 3000
# 206.8 -> 204.9 ms, batch 1 17.0 -> 16.6 ms (same box each, one capture stream per device): ON again.
WGRAD_STREAM = True
_wgrad_streams = {}


class _WeightGradLane:
    """Inside a backward: run the weight-gradient products on a second HIP stream beside the data-gradient chain.  A weight gradient and
    the data gradient of the same layer both only read the incoming gradient, so they are independent; issued back to back on one stream
    each of them ends in a tail of partly filled CUs (and a split-K fold launch) before the next may start - side by side the two kernels'
    workgroups interleave and fill each other's tails.  `run(fn)` issues fn() on the lane after everything launched so far; `join(*outs)`
    makes the launch stream wait for the lane and marks the tensors the lane allocated as used on the launch stream (caching-allocator
    hygiene).  Inputs of fn must stay alive and unmodified until join()."""

    def __init__(self, ref: torch.Tensor):
        # not under stream capture: a captured batch-1 step with the fork / join events of the lane in its graph replayed in 29.5 ms against 15.9 ms eager
        self.on = WGRAD_STREAM and ref.is_cuda and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.cur = torch.cuda.current_stream(ref.device)
            key = (ref.device, self.cur.cuda_stream)
            if key not in _wgrad_streams:
                _wgrad_streams[key] = torch.cuda.Stream(device=ref.device)
            self.lane = _wgrad_streams[key]

    def run(self, fn):
        if not self.on:
            return fn()
        self.lane.wait_event(self.cur.record_event())
        with torch.cuda.stream(self.lane):
            return fn()

    def join(self, *outs):
        if self.on:
            self.cur.wait_stream(self.lane)
            for t in outs:
                if t is not None:
                    t.record_stream(self.cur)


# ---- precision travels with the autograd node -----------------------------------------------------------------------------------
# `ops.backend().mm_precision` / `.attn_precision` select the product form (bf16x3 split = the parity setting, plain bf16 = 16-mixed, fp8
# attention forward).  A node's backward runs long after its forward, when the caller may have restored another setting (Faceformer.encode
# switches to bf16 around the encoder forward only): every forward records the setting it ran under and every backward re-instates it, so
# that e.g. the attention backward recomputes P = exp(q k^T - lse) with the product form the saved lse came from.
def _save_prec(ctx):
    be = ops.backend()
    ctx.prec = (be.mm_precision, be.attn_precision)


def _restores_precision(backward):
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        be = ops.backend()
        old = (be.mm_precision, be.attn_precision)
        be.mm_precision, be.attn_precision = ctx.prec
        try:
            return backward(ctx, *grads)
        finally:
            be.mm_precision, be.attn_precision = old
    return wrapped


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _new(ref, *shape):
    return torch.empty(*shape, device=ref.device, dtype=torch.float32)


# =================================================================================================
# building blocks
# =================================================================================================
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of (R, C) rows."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        be = ops.backend()
        xc, g, b = _f32c(x), _f32c(gamma), _f32c(beta)
        R, C = xc.shape
        y, mean, rstd = _new(xc, R, C), _new(xc, R), _new(xc, R)
        be.layernorm_fwd(xc, R, C, g, b, LN_EPS, y, mean, rstd)
        ctx.save_for_backward(xc, mean, rstd, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = ops.backend()
        xc, mean, rstd, g = ctx.saved_tensors
        R, C = xc.shape
        dx, dg, db = _new(xc, R, C), _new(xc, C), _new(xc, C)
        be.layernorm_bwd(_f32c(dy), xc, mean, rstd, g, R, C, dx, dg, db)
        return dx, dg, db


class LinearMMFn(torch.autograd.Function):
    """y = x W^T + b on the bf16-MFMA GEMM; x (M,K), W (N,K), b (N).  keep = (bits, scale): dropout of y in the epilogue."""

    @staticmethod
    def forward(ctx, x, W, b, keep=None):
        be = ops.backend()
        _save_prec(ctx)
        xc, Wc, bc = _f32c(x), _f32c(W), _f32c(b)
        M, K = xc.shape
        N = Wc.shape[0]
        y = _new(xc, M, N)
        be.mm(MMOp(xc, K), MMOp(Wc, K), y, M, N, K, N, bias=bc, keep=keep)
        ctx.save_for_backward(xc, Wc)
        ctx.keep = keep
        return y

    @staticmethod
    @_restores_precision
    def backward(ctx, dy):
        be = ops.backend()
        xc, Wc = ctx.saved_tensors
        M, K = xc.shape
        N = Wc.shape[0]
        dy = _f32c(dy)
        if ctx.keep is not None:
            dm = _new(dy, M, N)
            be.dropout_apply(dy, ctx.keep[0], M, N, N, ctx.keep[1], dm)
            dy = dm
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _new(xc, M, K)
            ops.mm_linear_dgrad(be, dy, Wc, dx, M, N, K)
        dW, db = _new(xc, N, K), _new(xc, N)
        ops.mm_linear_wgrad(be, dy, xc, dW, M, N, K, db=db)        # bias gradient collected inside the weight-gradient kernel
        return dx, dW, db, None


# =================================================================================================
# one post-norm Transformer layer (HF Wav2Vec2EncoderLayer): attention -> +res -> LN -> FFN(GELU) -> +res -> LN
# =================================================================================================
class EncoderLayerFn(torch.autograd.Function):
    """x (S*T, d) -> (S*T, d).  Parameters in HF order: q, k, v, out projections (W, b), layer_norm (g, b),
    intermediate_dense, output_dense (W, b), final_layer_norm (g, b).  19 launches forward, 31 backward.

    ``masks`` (train mode, else None): the keep bits of the layer's four dropout sites (HF Wav2Vec2EncoderLayer /
    Wav2Vec2FeedForward / eager attention): {"attn": (query-major, key-major, scale), "hid1": (bits, scale) on the out-projection
    output, "act": (bits, scale) on GELU(intermediate_dense), "hid2": (bits, scale) on the output_dense output}; absent keys = no
    dropout there.  Forward applies them inside the producing kernels (attention kernel, a2f_mm epilogues); backward masks the
    incoming gradient of the two hidden sites with one streaming pass each and the activation site in the data-gradient epilogue."""

    @staticmethod
    def use_images(be, M, d, F):
        """The image-operand path (csrc/gemm_img.hip): bf16x3 products only, widths in whole 32-blocks, enough rows to fill 256-row tiles."""
        return MM_IMG and be.mm_precision == 1 and d % 32 == 0 and F % 32 == 0 and M >= IMG_MIN_ROWS and hasattr(be, "split_img")

    @staticmethod
    def forward(ctx, x, x_img, S, T, heads, masks, Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, b1, W1, c1, W2, c2, g2, b2):
        """Returns (h2, h2_img): the layer output as fp32 rows and (image path; else None) as a split-32 image for the next layer's q|k|v product.
        x_img: the same for the input (None: made here by a splitting pass)."""
        be = ops.backend()
        _save_prec(ctx)
        xc = _f32c(x)
        ctx.wimg_owner = Wq                       # the module's parameter object: the weight-image cache entry lives on it (_layer_weight_images)
        P = [_f32c(p) for p in (Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, b1, W1, c1, W2, c2, g2, b2)]
        Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, b1, W1, c1, W2, c2, g2, b2 = P
        mk = masks or {}
        M, d = xc.shape
        F = W1.shape[0]
        keep = any(ctx.needs_input_grad)
        ctx.img = EncoderLayerFn.use_images(be, M, d, F)
        if ctx.img:
            return EncoderLayerFn._forward_img(ctx, be, xc, x_img, S, T, heads, mk, keep, Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, b1, W1, c1, W2, c2, g2, b2)
        # q | k | v as ONE product against the stacked (3d, d) weight (a 7 MB copy per call): one launch and one tile tail instead of three
        Wqkv, bqkv = torch.cat((Wq, Wk, Wv), 0), torch.cat((bq, bk, bv), 0)
        qkv = _new(xc, M, 3 * d)
        be.mm(MMOp(xc, d), MMOp(Wqkv, d), qkv, M, 3 * d, d, 3 * d, bias=bqkv)
        att, lse = _new(xc, M, d), _new(xc, S, heads, T)
        scale = float((d // heads) ** -0.5)
        if "attn" in mk:
            be.attn_fwd_drop(qkv, S, T, heads, 3 * d, scale, att, d, lse, mk["attn"][0], mk["attn"][2])
        else:
            be.attn_fwd(qkv, S, T, heads, 3 * d, scale, att, d, lse)
        a1 = _new(xc, M, d)
        be.mm(MMOp(att, d), MMOp(Wo, d), a1, M, d, d, d, bias=bo, aux=xc, ldaux=d, aux_mode=MM_AUX_ADD, keep=mk.get("hid1"))
        h1, m1, r1 = _new(xc, M, d), _new(xc, M), _new(xc, M)
        be.layernorm_fwd(a1, M, d, g1, b1, LN_EPS, h1, m1, r1)
        z = _new(xc, M, F) if keep else None                     # gelu'(pre-activation): what the backward multiplies by
        g = _new(xc, M, F)
        be.mm(MMOp(h1, d), MMOp(W1, d), g, M, F, d, F, bias=c1, act=MM_ACT_GELU_SAVE_GRAD if keep else MM_ACT_GELU, C2=z,
              keep=mk.get("act"))
        a2 = _new(xc, M, d)
        be.mm(MMOp(g, F), MMOp(W2, F), a2, M, d, F, d, bias=c2, aux=h1, ldaux=d, aux_mode=MM_AUX_ADD, keep=mk.get("hid2"))
        h2, m2, r2 = _new(xc, M, d), _new(xc, M), _new(xc, M)
        be.layernorm_fwd(a2, M, d, g2, b2, LN_EPS, h2, m2, r2)
        if keep:
            ctx.save_for_backward(xc, qkv, att, lse, a1, m1, r1, h1, z, g, a2, m2, r2, Wqkv, Wo, g1, W1, W2, g2)
            ctx.dims = (S, T, heads, scale)
            ctx.masks = mk
        return h2, None                         # no image on this path

    # ---- the same layer on split-32 image operands (csrc/gemm_img.hip): every GEMM input is an image written by its producer ----------------
    @staticmethod
    def _wimg(be, W):
        """(image of W, image of W^T): forward / weight-gradient products read W (N, K) k-contiguously, data gradients read W^T."""
        N, K = W.shape
        wi, wt = torch.empty_like(W), torch.empty(K, N, device=W.device, dtype=torch.float32)
        be.split_img(W, N, K, K, wi, K)
        be.split_img(W, N, K, K, wt, N, transpose=True)
        return wi, wt

    @staticmethod
    def _layer_weight_images(be, owner, Wq, Wk, Wv, bq, bk, bv, Wo, W1, W2):
        """The fused q|k|v weight / bias and the (image, transposed image) pairs of a layer's four weights.  They depend on the weights alone, and a step
        runs every layer several times on unchanged weights (the parts of the pipelined step): built once per (weights state, stream), 16 small launches per
        layer and part otherwise (1.35 ms of the 64 x 600 step).  The entry lives ON the layer's q_proj.weight parameter (`owner`: the tensor the caller's
        module holds), so it dies with the model and no process-wide table pins flat buffers of deleted models.  It is valid for one state of the weights: the
        addresses and tensor versions of all nine sources (in-place torch ops, load_state_dict) and `WEIGHTS_EPOCH`, which trainer.FlatAdam bumps (its Adam
        kernel writes the parameters through raw pointers).  A write through `p.data` or by a foreign kernel moves none of these - INTEGRATION.md 4; the debug
        guard `A2F_WIMG_VERIFY=1` re-splits on every hit and raises when the cached images are not what the weights give.  Never used under stream capture
        (a replay must run the splitting passes)."""
        def build():
            Wqkv, bqkv = torch.cat((Wq, Wk, Wv), 0), torch.cat((bq, bk, bv), 0)
            return (Wqkv, bqkv) + EncoderLayerFn._wimg(be, Wqkv) + EncoderLayerFn._wimg(be, Wo) + EncoderLayerFn._wimg(be, W1) + EncoderLayerFn._wimg(be, W2)
        if not WIMG_CACHE or not Wq.is_cuda or torch.cuda.is_current_stream_capturing():
            return build()
        src = (Wq, Wk, Wv, bq, bk, bv, Wo, W1, W2)
        tag = (WEIGHTS_EPOCH, torch.cuda.current_stream(Wq.device).cuda_stream) + tuple(t.data_ptr() for t in src) + tuple(t._version for t in src)
        ent = owner.__dict__.get("_a2f_wimg")
        if ent is not None and ent[0] == tag:
            if WIMG_VERIFY:
                fresh = build()
                for name, a, b in zip(("Wqkv", "bqkv", "Wqkv_i", "Wqkv_t", "Wo_i", "Wo_t", "W1_i", "W1_t", "W2_i", "W2_t"), ent[1], fresh):
                    if not torch.equal(a, b):
                        raise ops.A2FError(f"stale weight images ({name}): the layer's weights were rewritten behind torch's version counters (a write through "
                                           "`.data` or a foreign kernel) - call encoder_ops.bump_weights_epoch() after such a write")
            return ent[1]
        out = build()
        owner.__dict__["_a2f_wimg"] = (tag, out)
        return out

    @staticmethod
    def _forward_img(ctx, be, xc, x_img, S, T, heads, mk, keep, Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, b1, W1, c1, W2, c2, g2, b2):
        M, d = xc.shape
        F = W1.shape[0]
        I = lambda t, ld, **kw: MMOp(t, ld, img=True, **kw)
        if x_img is None:
            x_img = _new(xc, M, d)
            be.split_img(xc, M, d, d, x_img, d)
        Wqkv, bqkv, Wqkv_i, Wqkv_t, Wo_i, Wo_t, W1_i, W1_t, W2_i, W2_t = EncoderLayerFn._layer_weight_images(be, ctx.wimg_owner, Wq, Wk, Wv, bq, bk, bv, Wo, W1, W2)
        # q|k|v exists only as the image the attention kernels copy their tiles and fragments from (their inner loops are bound by vector instructions: converting
        # every K / V tile once per 128-query block was a fifth of them)
        ctx.attn_img = True                      # q|k|v and dctx travel as the images the projection GEMMs write
        qkv = _new(xc, M, 3 * d)
        be.mm(I(x_img, d), I(Wqkv_i, d), qkv, M, 3 * d, d, 3 * d, bias=bqkv, c_img=ctx.attn_img)
        att, att_img, lse = _new(xc, M, d), _new(xc, M, d), _new(xc, S, heads, T)
        scale = float((d // heads) ** -0.5)
        be.attn_fwd_img(qkv, S, T, heads, 3 * d, scale, att, d, lse, att_img, keep=(mk["attn"][0], mk["attn"][2]) if "attn" in mk else None, qkv_img=ctx.attn_img)
        a1 = _new(xc, M, d)
        be.mm(I(att_img, d), I(Wo_i, d), a1, M, d, d, d, bias=bo, aux=xc, ldaux=d, aux_mode=MM_AUX_ADD, keep=mk.get("hid1"))
        h1, h1_img, m1, r1 = _new(xc, M, d), _new(xc, M, d), _new(xc, M), _new(xc, M)
        be.layernorm_fwd_img(a1, M, d, g1, b1, LN_EPS, h1, h1_img, m1, r1)
        z = _new(xc, M, F) if keep else None
        g_img = _new(xc, M, F)                                  # GELU(intermediate_dense) exists only as the image the next product reads
        be.mm(I(h1_img, d), I(W1_i, d), g_img, M, F, d, F, bias=c1, act=MM_ACT_GELU_SAVE_GRAD if keep else MM_ACT_GELU, C2=z, keep=mk.get("act"), c_img=True)
        a2 = _new(xc, M, d)
        be.mm(I(g_img, F), I(W2_i, F), a2, M, d, F, d, bias=c2, aux=h1, ldaux=d, aux_mode=MM_AUX_ADD, keep=mk.get("hid2"))
        h2, h2_img, m2, r2 = _new(xc, M, d), _new(xc, M, d), _new(xc, M), _new(xc, M)
        be.layernorm_fwd_img(a2, M, d, g2, b2, LN_EPS, h2, h2_img, m2, r2)
        if keep:
            ctx.save_for_backward(x_img, qkv, att, att_img, lse, a1, m1, r1, h1_img, z, g_img, a2, m2, r2, Wqkv_t, Wo_t, W1_t, W2_t, g1, g2)
            ctx.dims = (S, T, heads, scale)
            ctx.masks = mk
        ctx.mark_non_differentiable(h2_img)
        return h2, h2_img

    @staticmethod
    def _backward_img(ctx, be, dh2):
        x_img, qkv, att, att_img, lse, a1, m1, r1, h1_img, z, g_img, a2, m2, r2, Wqkv_t, Wo_t, W1_t, W2_t, g1, g2 = ctx.saved_tensors
        S, T, heads, scale = ctx.dims
        mk = ctx.masks
        M, d = x_img.shape
        F = g_img.shape[1]
        new = lambda *s: _new(x_img, *s)
        I = lambda t, ld, **kw: MMOp(t, ld, img=True, **kw)
        dh2 = _f32c(dh2)
        lane = _WeightGradLane(x_img)
        # final_layer_norm: da2 (fp32: the residual branch) + the gradient output_dense receives, through its dropout mask, as an image + its column sums
        da2, dy2_img, dg2, db2, dc2 = new(M, d), new(M, d), new(d), new(d), new(d)
        be.layernorm_bwd_img(dh2, a2, m2, r2, g2, M, d, da2, dg2, db2, dxm_colsum=dc2, keep=mk.get("hid2"), dxm_img=dy2_img)
        dW2 = new(d, F)
        lane.run(lambda: be.mm(I(dy2_img, d, trans=True), I(g_img, F, trans=True), dW2, d, F, M, F, split_k=0))
        # through output_dense and GELU: the epilogue multiplies by the saved derivative (and the activation-dropout mask), writes the image the two
        # products of intermediate_dense read, and collects its column sums = intermediate_dense's bias gradient
        dz_img, dc1 = new(M, F), new(F)
        be.mm(I(dy2_img, d), I(W2_t, d), dz_img, M, F, d, F, keep=mk.get("act"), aux=z, aux_mode=MM_AUX_MUL_LIKE_C, c_img=True, c_colsum=dc1)
        dW1 = new(F, d)
        lane.run(lambda: be.mm(I(dz_img, F, trans=True), I(h1_img, d, trans=True), dW1, F, d, M, d, split_k=0))
        dh1 = new(M, d)
        be.mm(I(dz_img, F), I(W1_t, F), dh1, M, d, F, d, aux=da2, ldaux=d, aux_mode=MM_AUX_ADD)
        da1, dy1_img, dg1, db1, dbo = new(M, d), new(M, d), new(d), new(d), new(d)
        be.layernorm_bwd_img(dh1, a1, m1, r1, g1, M, d, da1, dg1, db1, dxm_colsum=dbo, keep=mk.get("hid1"), dxm_img=dy1_img)
        dWo = new(d, d)
        lane.run(lambda: be.mm(I(dy1_img, d, trans=True), I(att_img, d, trans=True), dWo, d, d, M, d, split_k=0))
        datt = new(M, d)
        be.mm(I(dy1_img, d), I(Wo_t, d), datt, M, d, d, d, c_img=ctx.attn_img)
        dqkv_img = new(M, 3 * d)
        be.attn_bwd_img(qkv, att, datt, lse, S, T, heads, 3 * d, d, scale, dqkv_img,
                        keep=(mk["attn"][0], mk["attn"][1], mk["attn"][2]) if "attn" in mk else None, inputs_img=ctx.attn_img)
        dbqkv, dWqkv = new(3 * d), new(3 * d, d)
        lane.run(lambda: (be.mm(I(dqkv_img, 3 * d, trans=True), I(x_img, d, trans=True), dWqkv, 3 * d, d, M, d, split_k=0),
                          be.colsum_img(dqkv_img, M, 3 * d, 3 * d, dbqkv)))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new(M, d)
            be.mm(I(dqkv_img, 3 * d), I(Wqkv_t, 3 * d), dx, M, d, 3 * d, d, aux=da1, ldaux=d, aux_mode=MM_AUX_ADD)
        lane.join()
        return (dx, None, None, None, None, None, dWqkv[0:d], dbqkv[0:d], dWqkv[d:2 * d], dbqkv[d:2 * d], dWqkv[2 * d:], dbqkv[2 * d:], dWo, dbo, dg1, db1,
                dW1, dc1, dW2, dc2, dg2, db2)

    @staticmethod
    @_restores_precision
    def backward(ctx, dh2, _dh2_img=None):
        be = ops.backend()
        if ctx.img:
            return EncoderLayerFn._backward_img(ctx, be, dh2)
        xc, qkv, att, lse, a1, m1, r1, h1, z, g, a2, m2, r2, Wqkv, Wo, g1, W1, W2, g2 = ctx.saved_tensors
        S, T, heads, scale = ctx.dims
        mk = ctx.masks
        M, d = xc.shape
        F = W1.shape[0]
        new = lambda *s: _new(xc, *s)
        dh2 = _f32c(dh2)
        # final_layer_norm; the column sums of its input gradient are output_dense's bias gradient
        da2, dg2, db2, dc2 = new(M, d), new(d), new(d), new(d)
        if "hid2" in mk:                 # the residual branch keeps da2; output_dense sees it through its dropout mask
            be.layernorm_bwd(dh2, a2, m2, r2, g2, M, d, da2, dg2, db2, None)
            dy2 = new(M, d)
            be.dropout_apply(da2, mk["hid2"][0], M, d, d, mk["hid2"][1], dy2)
        else:
            be.layernorm_bwd(dh2, a2, m2, r2, g2, M, d, da2, dg2, db2, dc2)
            dy2 = da2
        lane = _WeightGradLane(xc)
        dW2 = new(d, F)
        lane.run(lambda: ops.mm_linear_wgrad(be, dy2, g, dW2, M, d, F, db=dc2 if "hid2" in mk else None))
        dz = new(M, F)
        # through GELU: the forward epilogue saved gelu'(pre-activation), so this epilogue only multiplies (evaluating erf / exp here
        # per element dropped the K = 768 product from 245 to 122 TFLOP/s; a separate streaming GELU' pass cost 3 % of the step)
        ops.mm_linear_dgrad(be, dy2, W2, dz, M, d, F, keep=mk.get("act"), aux=z, aux_mode=MM_AUX_MUL_LIKE_C)
        dc1 = new(F)
        dW1 = new(F, d)
        lane.run(lambda: ops.mm_linear_wgrad(be, dz, h1, dW1, M, F, d, db=dc1))      # bias gradients ride along in the weight-gradient kernels

        dh1 = new(M, d)
        ops.mm_linear_dgrad(be, dz, W1, dh1, M, F, d, aux=da2, aux_mode=MM_AUX_ADD)           # + residual branch
        # layer_norm after attention; colsum of its input gradient = out_proj's bias gradient
        da1, dg1, db1, dbo = new(M, d), new(d), new(d), new(d)
        if "hid1" in mk:
            be.layernorm_bwd(dh1, a1, m1, r1, g1, M, d, da1, dg1, db1, None)
            dy1 = new(M, d)
            be.dropout_apply(da1, mk["hid1"][0], M, d, d, mk["hid1"][1], dy1)
        else:
            be.layernorm_bwd(dh1, a1, m1, r1, g1, M, d, da1, dg1, db1, dbo)
            dy1 = da1
        dWo = new(d, d)
        lane.run(lambda: ops.mm_linear_wgrad(be, dy1, att, dWo, M, d, d, db=dbo if "hid1" in mk else None))
        datt = new(M, d)
        ops.mm_linear_dgrad(be, dy1, Wo, datt, M, d, d)
        dqkv = new(M, 3 * d)
        if "attn" in mk:
            be.attn_bwd_drop(qkv, att, datt, lse, S, T, heads, 3 * d, d, scale, dqkv, mk["attn"][0], mk["attn"][1], mk["attn"][2])
        else:
            be.attn_bwd(qkv, att, datt, lse, S, T, heads, 3 * d, d, scale, dqkv)
        dbqkv, dWqkv = new(3 * d), new(3 * d, d)
        lane.run(lambda: ops.mm_linear_wgrad(be, dqkv, xc, dWqkv, M, 3 * d, d, db=dbqkv))     # the three projections' gradients in one product
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new(M, d)
            ops.mm_linear_dgrad(be, dqkv, Wqkv, dx, M, 3 * d, d, aux=da1, aux_mode=MM_AUX_ADD)   # K = 3d in one pass, + residual branch
        lane.join()                                  # outputs were allocated on the launch stream; the lane only wrote them
        return (dx, None, None, None, None, None, dWqkv[0:d], dbqkv[0:d], dWqkv[d:2 * d], dbqkv[d:2 * d], dWqkv[2 * d:], dbqkv[2 * d:], dWo, dbo, dg1, db1,
                dW1, dc1, dW2, dc2, dg2, db2)


class DropoutFn(torch.autograd.Function):
    """y = keep ? x * scale : 0 on (R, C) rows with bit-packed keep masks (csrc/dropout.hip); the backward is the same map."""

    @staticmethod
    def forward(ctx, x, bits, scale):
        be = ops.backend()
        xc = _f32c(x)
        R, C = xc.shape
        y = _new(xc, R, C)
        be.dropout_apply(xc, bits, R, C, C, scale, y)
        ctx.bits, ctx.scale = bits, scale
        return y

    @staticmethod
    def backward(ctx, dy):
        be = ops.backend()
        dyc = _f32c(dy)
        R, C = dyc.shape
        dx = _new(dyc, R, C)
        be.dropout_apply(dyc, ctx.bits, R, C, C, ctx.scale, dx)
        return dx, None, None


# =================================================================================================
# feature-encoder layer 0: conv1d(1 -> C, k = 10, stride 5) + GroupNorm(C groups) + GELU, channels-last output
# =================================================================================================
class FeatEnc0Fn(torch.autograd.Function):
    """wave (S, N), w (C, 1, 10), gamma (C), beta (C) -> (S, L, C), L = (N - 10) // 5 + 1   (csrc/featenc.hip)."""

    @staticmethod
    def forward(ctx, wave, w, gamma, beta, out_img=False):
        """out_img: the returned tensor holds the split-32 IMAGE of the layer output (its only reader: ConvStackFn on image operands)."""
        be = ops.backend()
        _save_prec(ctx)
        xc, wc, g, b = _f32c(wave), _f32c(w), _f32c(gamma), _f32c(beta)
        S, N = xc.shape
        C = wc.shape[0]
        L = (N - 10) // 5 + 1
        out, stats = _new(xc, S, L, C), _new(xc, S, 2, C)
        if out_img:
            be.featenc0_fwd(xc, S, N, wc, g, b, C, 1e-5, out, stats, out_img=True)
        else:
            be.featenc0_fwd(xc, S, N, wc, g, b, C, 1e-5, out, stats)
        ctx.save_for_backward(xc, wc, g, b, stats)
        return out

    @staticmethod
    @_restores_precision
    def backward(ctx, dout):
        be = ops.backend()
        xc, wc, g, b, stats = ctx.saved_tensors
        S, N = xc.shape
        C = wc.shape[0]
        L = (N - 10) // 5 + 1
        need_dx = ctx.needs_input_grad[0]
        dw, dg, db = _new(xc, C, 1, 10), _new(xc, C), _new(xc, C)
        dy = _new(xc, S, L, C) if need_dx else None
        be.featenc0_bwd(xc, S, N, wc, g, b, C, stats, _f32c(dout), dw, dg, db, dy)
        dx = None
        if need_dx:
            # d/d(waveform) (only when the input itself requires a gradient - parity tests; training never asks for it):
            # dcol[row, k] = sum_c dy[row, c] w[c, k] on the GEMM (N padded 10 -> 12), then the 10 strided taps are added up
            wp = torch.zeros(12, C, device=xc.device, dtype=torch.float32)
            wp[:10] = wc.view(C, 10).t()
            dcol = _new(xc, S * L, 12)
            be.mm(MMOp(dy, C), MMOp(wp, C), dcol, S * L, 12, C, 12)
            dcol = dcol.view(S, L, 12)
            dx = torch.zeros(S, N, device=xc.device, dtype=torch.float32)
            for k in range(10):
                dx[:, k:k + 5 * (L - 1) + 1:5] += dcol[:, :, k]
        return dx, dw, dg, db, None


# =================================================================================================
# channels-last Conv1d + GELU (feature-encoder layers 1..6: bias-free, k = 3 or 2, stride 2)
# =================================================================================================
class ConvStackFn(torch.autograd.Function):
    """Feature-encoder layers 1..n in one autograd node: x (S, L0, C0) -> gelu(conv_n(... gelu(conv_1(x)))) channels-last; weights are
    the nn.Conv1d weights (Co, Ci, KH), no bias, no padding; ``strides`` a tuple.  Forward epilogues save gelu'(pre-activation) of every
    layer; in the backward the data-gradient epilogue of layer i + 1 multiplies by layer i's saved derivative, so the gradient arrives at
    layer i already as its pre-activation gradient - one streaming GELU' multiply (the top layer's) instead of one per layer."""

    @staticmethod
    def use_images(be, x, weights):
        """Image operands (csrc/gemm_img.hip) for the whole stack: bf16x3 products, channel counts in whole 32-blocks, enough rows, every operand and
        output span below 2 GiB (32-bit epilogue offsets)."""
        S, L0, C0 = x.shape
        return (MM_IMG and be.mm_precision == 1 and hasattr(be, "split_img") and C0 % 32 == 0 and all(w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 for w in weights)
                and S * L0 >= 4 * IMG_MIN_ROWS and S * L0 * C0 * 4 < (1 << 31))

    @staticmethod
    def forward(ctx, x, strides, x_is_img, *weights):
        """x_is_img: x already holds the split-32 image of the stack's input (FeatEnc0Fn with out_img)."""
        be = ops.backend()
        _save_prec(ctx)
        cur = _f32c(x)
        keep = any(ctx.needs_input_grad)
        ctx.img = bool(x_is_img) or ConvStackFn.use_images(be, cur, weights)
        if ctx.img:
            return ConvStackFn._forward_img(ctx, be, cur, bool(x_is_img), strides, keep, weights)
        saved, dims = [], []
        for w, stride in zip(weights, strides):
            wc = _f32c(w)
            S, Lin, Ci = cur.shape
            Co, _, KH = wc.shape
            Lout = (Lin - KH) // stride + 1
            wg = _new(cur, Co, KH * Ci)
            be.conv_weight_pack(wc, wg, Co, Ci, KH, 0)
            gp = _new(cur, S, Lout, Co) if keep else None
            y = _new(cur, S, Lout, Co)
            ops.mm_conv_fwd(be, cur, wg, y, S, Lin, Lout, Ci, Co, KH, stride, act=MM_ACT_GELU_SAVE_GRAD if keep else MM_ACT_GELU, pre=gp)
            saved += [cur, wg, gp]
            dims.append((S, Lin, Lout, Ci, Co, KH, stride))
            cur = y
        if keep:
            ctx.save_for_backward(*saved)
            ctx.dims = dims
        return cur

    # ---- the stack on image operands: every activation between the layers exists only as a split-32 image written by the producing epilogue ----------
    @staticmethod
    def _forward_img(ctx, be, cur, x_is_img, strides, keep, weights):
        I = lambda t, ld, **kw: MMOp(t, ld, img=True, **kw)
        if not x_is_img:
            ci = torch.empty_like(cur)
            be.split_img(cur, cur.shape[0] * cur.shape[1], cur.shape[2], cur.shape[2], ci, cur.shape[2])
            cur = ci
        saved, dims = [], []
        n = len(weights)
        for li, (w, stride) in enumerate(zip(weights, strides)):
            wc = _f32c(w)
            S, Lin, Ci = cur.shape
            Co, _, KH = wc.shape
            Lout = (Lin - KH) // stride + 1
            wg, wgi = _new(cur, Co, KH * Ci), _new(cur, Co, KH * Ci)
            be.conv_weight_pack(wc, wg, Co, Ci, KH, 0)
            be.split_img(wg, Co, KH * Ci, KH * Ci, wgi, KH * Ci)
            gp = _new(cur, S, Lout, Co) if keep else None
            y = _new(cur, S, Lout, Co)
            last = li == n - 1                                     # the top layer's output goes on to the interpolation as fp32 rows
            A = I(cur, stride * Ci, lines_per_seq=Lout, seq_stride=Lin * Ci, seq_elems=Lin * Ci)
            be.mm(A, I(wgi, KH * Ci), y, S * Lout, Co, KH * Ci, Co, act=MM_ACT_GELU_SAVE_GRAD if keep else MM_ACT_GELU, C2=gp, c_img=not last)
            saved += [cur, wg, gp]
            dims.append((S, Lin, Lout, Ci, Co, KH, stride))
            cur = y
        if keep:
            ctx.save_for_backward(*saved)
            ctx.dims = dims
        return cur

    @staticmethod
    def _backward_img(ctx, be, dy):
        saved, dims = ctx.saved_tensors, ctx.dims
        n = len(dims)
        I = lambda t, ld, **kw: MMOp(t, ld, img=True, **kw)

        def img_of(t2d):
            o = torch.empty_like(t2d)
            be.split_img(t2d, t2d.shape[0], t2d.shape[1], t2d.shape[1], o, t2d.shape[1])
            return o
        dzf = (_f32c(dy) * saved[3 * (n - 1) + 2])                # top layer: dy o gelu' (small: 49 Hz frames)
        dz = img_of(dzf.view(-1, dzf.shape[-1])).view(dzf.shape)
        dws = [None] * n
        dx = None
        lane, held = _WeightGradLane(dz), []
        for i in range(n - 1, -1, -1):
            xi, wg, _gp = saved[3 * i: 3 * i + 3]
            S, Lin, Lout, Ci, Co, KH, stride = dims[i]
            dwg, dw = _new(xi, Co, KH * Ci), _new(xi, Co, Ci, KH)

            def wgrad(xi=xi, dz=dz, dwg=dwg, dw=dw, dm=(S, Lin, Lout, Ci, Co, KH, stride)):
                S_, Lin_, Lout_, Ci_, Co_, KH_, st_ = dm
                B = I(xi, st_ * Ci_, trans=True, lines_per_seq=Lout_, seq_stride=Lin_ * Ci_, seq_elems=Lin_ * Ci_)
                be.mm(I(dz, Co_, trans=True), B, dwg, Co_, KH_ * Ci_, S_ * Lout_, KH_ * Ci_, split_k=0)
                be.conv_weight_pack(dwg, dw, Co_, Ci_, KH_, 1)
            lane.run(wgrad)
            held.append((dz, dwg))
            dws[i] = dw
            if i == 0 and not ctx.needs_input_grad[0]:
                break
            # data gradient in gather form; below layer 1 it lands as the image the next (lower) layer's two gradient products read, already multiplied
            # by that layer's saved gelu' - the bottom one as fp32 rows for layer 0's backward
            covered = Lout * KH if KH == stride else 2 * Lout + 1
            dxi = torch.empty(S, Lin, Ci, device=xi.device, dtype=torch.float32)
            if covered < Lin:
                dxi[:, covered:].zero_()
            ex = dict(aux=saved[3 * (i - 1) + 2], aux_mode=MM_AUX_MUL_LIKE_C) if i > 0 else {}
            cimg = i > 0
            if KH == stride:
                wt = img_of(wg.t().contiguous())                                       # (KH * Ci, Co)
                be.mm(I(dz, Co), I(wt, Co), dxi, S * Lout, KH * Ci, Co, KH * Ci, c_rows_per_seq=Lout, c_seq_stride=Lin * Ci, c_img=cimg, **ex)
            else:
                assert KH == 3 and stride == 2 and Lin >= 2 * Lout + 1
                rows = Lout + 1
                # positions 2 q: dy[q - 1] through tap 2 + dy[q] through tap 0 - a two-row window of dz (zero rows beyond both ends) against [W_2 ; W_0]
                w20 = img_of(torch.cat((wg[:, 2 * Ci:3 * Ci].t(), wg[:, 0:Ci].t()), 1).contiguous())     # (Ci, 2 Co)
                A = I(dz, Co, lines_per_seq=rows, seq_stride=Lout * Co, off=-Co, seq_elems=Lout * Co)
                be.mm(A, I(w20, 2 * Co), dxi, S * rows, Ci, 2 * Co, 2 * Ci, c_off=0, c_rows_per_seq=rows, c_seq_stride=Lin * Ci, c_img=cimg, **ex)
                # positions 2 q + 1: dy[q] through the middle tap
                w1 = img_of(wg[:, Ci:2 * Ci].t().contiguous())                                             # (Ci, Co)
                be.mm(I(dz, Co), I(w1, Co), dxi, S * Lout, Ci, Co, 2 * Ci, c_off=Ci, c_rows_per_seq=Lout, c_seq_stride=Lin * Ci, c_img=cimg, **ex)
            if i > 0:
                dz = dxi
            else:
                dx = dxi
        lane.join()
        return (dx, None, None, *dws)

    @staticmethod
    @_restores_precision
    def backward(ctx, dy):
        be = ops.backend()
        if ctx.img:
            return ConvStackFn._backward_img(ctx, be, dy)
        saved, dims = ctx.saved_tensors, ctx.dims
        n = len(dims)
        dz = _f32c(dy) * saved[3 * (n - 1) + 2]                   # top layer: dy o gelu'
        dws = [None] * n
        dx = None
        lane, held = _WeightGradLane(dz), []
        for i in range(n - 1, -1, -1):
            xc, wg, _gp = saved[3 * i: 3 * i + 3]
            S, Lin, Lout, Ci, Co, KH, stride = dims[i]
            dwg, dw = _new(xc, Co, KH * Ci), _new(xc, Co, Ci, KH)

            def wgrad(xc=xc, dz=dz, dwg=dwg, dw=dw, dm=(S, Lin, Lout, Ci, Co, KH, stride)):
                ops.mm_conv_wgrad(be, xc, dz, dwg, *dm)
                be.conv_weight_pack(dwg, dw, dm[4], dm[3], dm[5], 1)
            lane.run(wgrad)
            held.append((dz, dwg))                                 # read / written by the lane: alive until join()
            dws[i] = dw
            if i == 0 and not ctx.needs_input_grad[0]:
                break
            # the gather-form data gradient writes positions [0, covered); only the tail no window reaches needs zeros (a full fill of
            # the layer-1 gradient is 2 GB per half-batch)
            covered = Lout * KH if KH == stride else 2 * Lout + 1
            dxi = torch.empty(S, Lin, Ci, device=xc.device, dtype=torch.float32)
            if covered < Lin:
                dxi[:, covered:].zero_()
            ops.mm_conv_dgrad(be, dz, wg, dxi, S, Lin, Lout, Ci, Co, KH, stride, mul=saved[3 * (i - 1) + 2] if i > 0 else None)
            if i > 0:
                dz = dxi                                           # already layer (i - 1)'s pre-activation gradient
            else:
                dx = dxi
        lane.join()
        return (dx, None, None, *dws)


# =================================================================================================
# grouped positional convolution (HF Wav2Vec2PositionalConvEmbedding): h + gelu(conv1d(h, k=128, pad=64, groups=16)[:-1])
# =================================================================================================
class PosConvFn(torch.autograd.Function):
    """h (S, T, C) , w (C, C/groups, K) (weight-norm already applied), bias (C) -> h + gelu(grouped conv)  (S, T, C) on the kernels of csrc/posconv.hip: the frame
    block of a group sits in LDS once for all 128 taps, the activation is read in place (left pad K/2, even kernel: the last output frame is dropped).  wav2vec2-base's
    shape only (16 groups x 48 channels x 128 taps) - the shape the reference's encoder has (wav2vec.py:87-109)."""

    @staticmethod
    def forward(ctx, h, w, bias, groups):
        be = ops.backend()
        _save_prec(ctx)
        hc, wc, bc = _f32c(h), _f32c(w), _f32c(bias)
        S, T, C = hc.shape
        Cg, K = wc.shape[1], wc.shape[2]
        G = groups
        pad = K // 2
        if (G, Cg, K) != (16, 48, 128):
            raise ops.A2FError(f"positional convolution: the HIP kernels are built for 16 groups x 48 channels x 128 taps (wav2vec2-base); got {G} x {Cg} x {K}")
        keep = any(ctx.needs_input_grad)
        wimg = _new(hc, C, K * Cg)
        be.split_img(wc.view(G, Cg, Cg, K).permute(0, 1, 3, 2).contiguous().view(C, K * Cg), C, K * Cg, K * Cg, wimg, K * Cg)     # [g*Cg+co][kk*Cg+ci]
        z = _new(hc, S * T, C) if keep else None
        y = _new(hc, S * T, C)
        be.posconv_nt(hc, wimg, y, S, T, C, G, K, -pad, bias=bc, aux=hc, z=z, act=1)
        if keep:
            ctx.save_for_backward(hc, wc, z)
            ctx.dims = (S, T, C, Cg, K, G)
        return y.view(S, T, C)

    @staticmethod
    @_restores_precision
    def backward(ctx, dy):
        be = ops.backend()
        hc, wc, z = ctx.saved_tensors
        S, T, C, Cg, K, G = ctx.dims
        pad = K // 2
        dy = _f32c(dy).view(S * T, C)
        dz = _new(dy, S * T, C)
        be.gelu_bwd(dy, z, dz)
        db = _new(dy, C)
        be.colsum(dz, S * T, C, C, db)
        dwg = _new(dy, G, K, Cg, Cg)                                          # [g][kk][ci][co]
        be.posconv_wgrad(hc, dz, dwg, S, T, C, G, K, pad)
        dw = dwg.permute(0, 3, 2, 1).reshape(C, Cg, K)                        # -> (g*Cg + co, ci, kk)
        dh = None
        if ctx.needs_input_grad[0]:
            # dh[s,t',g,ci] = sum_{kk',co} dz[s, t'-(K/2-1)+kk', g, co] * w[g*Cg+co, ci, K-1-kk']  (+ the residual branch)
            wd = wc.view(G, Cg, Cg, K).flip(-1).permute(0, 2, 3, 1).contiguous().view(C, K * Cg)       # [g*Cg+ci][kk'*Cg+co]
            wdimg = _new(dy, C, K * Cg)
            be.split_img(wd, C, K * Cg, K * Cg, wdimg, K * Cg)
            dh = _new(dy, S * T, C)
            be.posconv_nt(dz, wdimg, dh, S, T, C, G, K, -(pad - 1), aux=dy)
            dh = dh.view(S, T, C)
        return dh, dw, db, None
