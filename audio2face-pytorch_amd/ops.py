"""Host-side launch layer: torch tensors in, HIP kernels (liba2f_hip.so, C ABI) out.

``HipBackend`` is the only backend the product ever instantiates; every method is a 1:1 call of an
``a2f_*`` entry point on the current torch HIP stream.  The layer composers below (implicit-GEMM
convolutions, Linear forward/backward) are written against the backend *interface* so that the
CPU test-suite can check their index arithmetic with an emulator injected through ``set_backend``
(tests only; the package itself never does that and has no CPU path).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import A2FError, GemmDesc, MMDesc, MMOperand

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
AUX_NONE, AUX_RELU_BWD, AUX_TANH_BWD, AUX_ADD = 0, 1, 2, 3
# a2f_mm (bf16-MFMA GEMM family) epilogue codes
MM_ACT_NONE, MM_ACT_RELU, MM_ACT_GELU, MM_ACT_GELU_SAVE_GRAD = 0, 1, 3, 4
MM_AUX_NONE, MM_AUX_ADD, MM_AUX_GELU_BWD, MM_AUX_RELU_BWD, MM_AUX_MUL_LIKE_C = 0, 1, 2, 3, 4


class MMOp:
    """One operand of ``HipBackend.mm`` (a2f_mm_operand): tensor + line stride; ``trans`` = the contiguous axis is m / n
    instead of k; ``lines_per_seq`` > 0 switches on the window structure (sequence stride, in-sequence offset, bounds)."""

    def __init__(self, t: torch.Tensor, ld: int, *, trans: bool = False, lines_per_seq: int = 0, seq_stride: int = 0, off: int = 0,
                 seq_elems: int = 0, ptr_off: int = 0, img: bool = False):
        self.t, self.ld, self.trans = t, int(ld), bool(trans)
        self.img = bool(img)            # the tensor holds a split-32 image (a2f_mm_operand.format 1) of the fp32 matrix of the same shape
        self.lines_per_seq, self.seq_stride, self.off, self.seq_elems, self.ptr_off = int(lines_per_seq), int(seq_stride), int(off), int(seq_elems), int(ptr_off)

    def fill(self, o: MMOperand):
        o.p = self.t.data_ptr() + 4 * self.ptr_off
        o.bytes = 4 * (self.t.numel() - self.ptr_off)
        o.ld, o.trans = self.ld, int(self.trans)
        o.format = 1 if self.img else 0
        o.lines_per_seq, o.seq_stride, o.off, o.seq_elems = self.lines_per_seq, self.seq_stride, self.off, self.seq_elems


_raw_stream = torch._C._cuda_getCurrentRawStream
_TLS = threading.local()


def _ptr(t: Optional[torch.Tensor], off: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + 4 * off


_F32 = torch.float32


def _chk(*ts):
    # (attribute reads only: `t.device.type` builds a torch.device per call - this guard runs ~360 times per batch-1 FaceFormer step, which the host's launch rate bounds)
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise A2FError("a2f ops need HIP (cuda) tensors; this package has no CPU path")
        if t.dtype is not _F32:
            raise A2FError(f"a2f ops compute in fp32; got {t.dtype}")
        if not t.is_contiguous():
            raise A2FError("a2f ops need contiguous tensors")


SMALL_M_SPLIT = 1024      # products of at most this many rows are cut along K automatically (module attribute; 0: never)


def _chk_bits(*ts):
    for t in ts:
        if not t.is_cuda or t.dtype is not torch.int32 or not t.is_contiguous():
            raise A2FError("keep-bit masks are contiguous int32 HIP tensors (32 elements per word)")


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self.prof = None          # when a dict: tag -> list of (start_event, end_event) recorded on the launch stream
        self.auto_split_k = True  # gemm(): slab split-K for under-filled grids (see gemm)
        self.gemm_precision = 0   # a2f_gemm_desc.precision of subsequent gemm() calls: 0 exact fp32 MFMA, 1 bf16x3 split products
        self.wgrad_precision = 0  # a2f_gemm_desc.precision of subsequent wgrad_item() calls: 0 exact fp32 MFMA, 1 bf16x3

    @staticmethod
    def _stream(t: torch.Tensor):
        # the raw handle of torch's current stream on the tensor's device, straight from the C side: torch.cuda.current_stream(...).cuda_stream builds a Python
        # Stream object per call (~4 us, 170 times per batch-1 FaceFormer step, on a step that is bound by the host's launch rate: tools/host_bound_probe.py)
        return C.c_void_p(_raw_stream(t.get_device()))

    # ---- optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ----
    def _prof_begin(self, tag):
        if self.prof is None or tag is None:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def _prof_end(self, tag, ev):
        if ev is not None:
            ev[1].record()
            self.prof.setdefault(tag, []).append(ev)

    def _prof_flops(self, tag, ev, flops, key):
        """End a profiled launch (see _prof_begin) and book its flops under `tag` / its shape key (bench.py's gemm_family and per-shape tables)."""
        self._prof_end(tag, ev)
        if self.prof is not None and tag is not None:
            pf = getattr(self, "prof_flops", None)
            if pf is not None:
                pf[tag] = pf.get(tag, 0.0) + flops
            ps = getattr(self, "prof_shapes", None)
            if ps is not None and ev is not None:
                ps.setdefault(key, {"events": [], "flops": flops})["events"].append(ev)

    # ---- a2f_gemm_f32 -------------------------------------------------------------------------
    def gemm(self, A, B, Cout, M, N, K, a: Tuple[int, int, int], b: Tuple[int, int, int], ldc, *, c_off=0, bias=None,
             act=ACT_NONE, aux=None, aux_off=0, ldaux=0, aux_mode=AUX_NONE, accumulate=False, split_k=1, ones_out=None,
             mask: Optional[Tuple[int, int, int, int, int]] = None, tag: Optional[str] = None, splitk_ws=None):
        _chk(A, B, Cout, bias, aux, ones_out, splitk_ws)
        if split_k == 1 and self.auto_split_k and K >= 256:
            # few output tiles and a long contraction (e.g. the Audio2Mesh articulation convolutions: 1..64 tiles, K = 768):
            # deterministic slab split-K spreads the k-loop over the chip; the fold applies the epilogue
            neff = N + (1 if ones_out is not None else 0)
            tiles = ((M + 127) // 128) * ((neff + 127) // 128)
            if tiles < 96:
                split_k = max(1, min(256 // tiles, K // 128, 16))
                if split_k > 1:
                    splitk_ws = torch.empty(split_k * M * (N + 1), device=Cout.device, dtype=torch.float32)
        d = GemmDesc()
        d.A, d.a_off, d.a_rs, d.a_cs = _ptr(A), a[0], a[1], a[2]
        d.B, d.b_off, d.b_rs, d.b_cs = _ptr(B), b[0], b[1], b[2]
        d.C, d.ldc = _ptr(Cout, c_off), ldc
        d.M, d.N, d.K = M, N, K
        d.bias = _ptr(bias)
        d.aux, d.ldaux = _ptr(aux, aux_off), ldaux
        d.ones_out = _ptr(ones_out)
        d.act, d.aux_mode, d.accumulate, d.split_k = act, aux_mode, int(accumulate), split_k
        d.precision = self.gemm_precision
        if mask is not None:
            d.mask_operand, d.mask_period, d.mask_phase, d.mask_lo, d.mask_hi = mask
        if splitk_ws is not None:
            if splitk_ws.numel() < split_k * M * (N + 1):
                raise A2FError("gemm: split-K workspace too small")
            d.splitk_ws = _ptr(splitk_ws)
        ev = self._prof_begin(tag)
        _lib.check(self.lib.a2f_gemm_f32(C.byref(d), self._stream(Cout)), "a2f_gemm_f32")
        self._prof_end(tag, ev)

    # ---- a2f_mm (bf16-MFMA GEMM family) ---------------------------------------------------------
    mm_precision = 1          # 1 = bf16x3 split products (fp32-class, the parity setting); 0 = plain bf16 operands (16-mixed)
    attn_precision = None     # None: the attention kernels follow mm_precision; 2 = single-term fp8 (e4m3) forward products, bf16 backward (a labelled
                              # variant: 1.3e-3 .. 3.8e-2 on T = 3000 vertex offsets - NOT a setting of a 1e-3 configuration)
    attn_precisions = (0, 1, 2)

    def mm(self, A: "MMOp", B: "MMOp", Cout, M, N, K, ldc, *, c_off=0, c_rows_per_seq=0, c_seq_stride=0, C2=None, bias=None, aux=None,
           ldaux=0, act=MM_ACT_NONE, aux_mode=MM_AUX_NONE, precision=None, split_k=1, batch=1, a_bstride=0, b_bstride=0, c_bstride=0,
           bias_bstride=0, aux_bstride=0, accumulate=False, alpha=1.0, tag="gemm_mfma", keep=None, a_rowsum=None, c_img=False, c_colsum=None):
        """C[m,n] = epi(alpha * sum_k A(m,k) B(k,n)) on the bf16 matrix pipe (contract: include/a2f_hip.h, a2f_mm).
        keep = (bits int32 (M, wpr), scale): train-mode dropout of C in the epilogue.  c_img: C is written as a split-32 image.
        c_colsum (N): receives the column sums of the stored C (image operands only)."""
        _chk(A.t, B.t, Cout, C2, bias, aux)
        prec = self.mm_precision if precision is None else precision
        if split_k == 0 and A.img:                              # image operands: 256 x 256 tiles, one workgroup per CU
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            split_k = max(1, min(256 // max(tiles, 1), K // 512, 128))
        elif split_k == 0:                                      # auto: enough workgroups for the chip, slabs stay small
            tiles = ((M + 127) // 128) * ((N + (127 if N > 64 else 63)) // (128 if N > 64 else 64)) * batch
            # (a one- or two-tile weight gradient - Audio2Mesh's analysis layer 1: 108 x 216 over 524 288 rows - needs more than 64 cuts to put a workgroup on every CU)
            split_k = max(1, min(512 // max(tiles, 1), K // 256, 64 if tiles >= 4 else 256 // max(tiles, 1))) if tiles < 192 else 1
        elif (split_k == 1 and SMALL_M_SPLIT and batch == 1 and not A.img and not A.trans and not accumulate and a_rowsum is None and c_colsum is None
              and M <= SMALL_M_SPLIT and N > 64 and K >= 384):
            # a product of few rows (the reference's own batch-1 FaceFormer step: 600 rows) makes (M / 128) (N / 128) workgroups - 30 for N = 768 - that
            # each walk the whole K alone: cut K so that ~256 workgroups run three or more 32-deep k-tiles each; the fold applies the epilogue
            tiles = ((M + 127) // 128) * ((N + 127) // 128)
            if tiles <= 128:
                split_k = max(1, min(256 // tiles, K // 96, 32))
        if keep is not None:
            bits, kscale = keep
            _chk_bits(bits)
            kwpr, kscale = bits.shape[-1], float(kscale)
        else:
            bits, kwpr, kscale = None, 0, 0.0
        # Everything of the descriptor that is not an address is the same at every step for a given call site: the filled struct is kept under the tuple of those
        # values and a call rewrites the dozen pointer / extent fields only (a ctypes field store costs ~0.2 us and the struct has 60 of them; a batch-1 FaceFormer
        # step makes 157 products and is bound by the host's launch rate: tools/host_bound_probe.py).
        key = (A.ld, A.trans, A.img, A.lines_per_seq, A.seq_stride, A.off, A.seq_elems, B.ld, B.trans, B.img, B.lines_per_seq, B.seq_stride, B.off, B.seq_elems,
               M, N, K, ldc, c_rows_per_seq, c_seq_stride, ldaux, act, aux_mode, prec, c_img, split_k, batch, a_bstride, b_bstride, c_bstride, bias_bstride,
               aux_bstride, accumulate, alpha, kwpr, kscale)
        cache = getattr(_TLS, "mm_descs", None)                 # per thread: the autograd engine calls in from its own thread
        if cache is None:
            cache = _TLS.mm_descs = {}
        d = cache.get(key)
        if d is None:
            if len(cache) > 4096:
                cache.clear()
            d = cache[key] = MMDesc()
            A.fill(d.A)
            B.fill(d.B)
            d.ldc, d.c_rows_per_seq, d.c_seq_stride = ldc, c_rows_per_seq, c_seq_stride
            d.M, d.N, d.K = M, N, K
            d.ldaux = ldaux
            d.act, d.aux_mode = act, aux_mode
            d.precision = prec
            d.c_format = 1 if c_img else 0
            d.split_k = split_k
            d.batch, d.accumulate = batch, int(accumulate)
            d.a_bstride, d.b_bstride, d.c_bstride, d.bias_bstride, d.aux_bstride = a_bstride, b_bstride, c_bstride, bias_bstride, aux_bstride
            d.alpha = alpha
            d.keep_wpr, d.keep_scale = kwpr, kscale
        ta, tb = A.t, B.t
        d.A.p, d.A.bytes = ta.data_ptr() + 4 * A.ptr_off, 4 * (ta.numel() - A.ptr_off)
        d.B.p, d.B.bytes = tb.data_ptr() + 4 * B.ptr_off, 4 * (tb.numel() - B.ptr_off)
        d.C = _ptr(Cout, c_off)
        d.C2 = _ptr(C2, c_off)
        d.bias = _ptr(bias)
        d.aux = _ptr(aux, c_off if aux_mode == MM_AUX_MUL_LIKE_C else 0)
        d.keep_bits = bits.data_ptr() if bits is not None else None
        ws = None
        if split_k > 1:
            ws = torch.empty(batch * split_k * M * N + (split_k * M if a_rowsum is not None else 0), device=Cout.device, dtype=torch.float32)
        if c_colsum is not None:
            _chk(c_colsum)
            ws = torch.empty(2 * ((M + 255) // 256) * N, device=Cout.device, dtype=torch.float32)
        d.ws = _ptr(ws)
        d.c_colsum = _ptr(c_colsum)
        if a_rowsum is not None:
            _chk(a_rowsum)
        d.a_rowsum = _ptr(a_rowsum)
        ev = self._prof_begin(tag)
        _lib.check(self.lib.a2f_mm(C.byref(d), self._stream(Cout)), "a2f_mm")
        self._prof_end(tag, ev)
        if self.prof is not None and tag is not None:
            pf = getattr(self, "prof_flops", None)
            if pf is not None:
                pf[tag] = pf.get(tag, 0.0) + 2.0 * M * N * K * batch
            ps = getattr(self, "prof_shapes", None)
            if ps is not None and ev is not None:                # per launch variant: layout / shape / precision -> events, flops per launch
                key = (("T" if A.trans else "N") + ("T" if not B.trans else "N"), M, N, K, batch, bool(A.lines_per_seq or B.lines_per_seq),
                       d.precision, bool(A.img))
                ent = ps.setdefault(key, {"events": [], "flops": 2.0 * M * N * K * batch})
                ent["events"].append(ev)

    def split_img(self, x, R, Cc, ld, out, ldo, transpose=False):
        """fp32 rows -> split-32 image (a2f_split_img); transpose: the image of x^T."""
        _chk(x, out)
        _lib.check(self.lib.a2f_split_img(_ptr(x), R, Cc, ld, int(transpose), _ptr(out), ldo, self._stream(x)), "a2f_split_img")

    # ---- grouped positional convolution (csrc/posconv.hip) ----------------------------------------
    def posconv_nt(self, x, w_img, y, S, T, Cc, groups, taps, off, *, bias=None, aux=None, z=None, act=0, tag="gemm_mfma"):
        """y = [aux +] act(bias + grouped window product of x (S,T,C) with the weight image w_img (C, taps*C/groups)) (a2f_posconv_nt)."""
        _chk(x, w_img, y, bias, aux, z)
        ev = self._prof_begin(tag)
        _lib.check(self.lib.a2f_posconv_nt(_ptr(x), _ptr(w_img), _ptr(bias) if bias is not None else None, _ptr(aux) if aux is not None else None, _ptr(y),
                                           _ptr(z) if z is not None else None, S, T, Cc, groups, taps, off, int(act), self._stream(x)), "a2f_posconv_nt")
        self._prof_flops(tag, ev, 2.0 * S * T * Cc * taps * (Cc // groups), ("NT", S * T, Cc // groups, taps * (Cc // groups), groups, True, 1, False))

    def posconv_wgrad(self, x, dz, dw, S, T, Cc, groups, taps, pad, tag="gemm_mfma"):
        """dw (G, taps, C/G, C/G) [g][tap][ci][co] = sum over frames of x windows x dz (a2f_posconv_wgrad)."""
        _chk(x, dz, dw)
        ev = self._prof_begin(tag)
        _lib.check(self.lib.a2f_posconv_wgrad(_ptr(x), _ptr(dz), _ptr(dw), S, T, Cc, groups, taps, pad, self._stream(x)), "a2f_posconv_wgrad")
        self._prof_flops(tag, ev, 2.0 * S * T * Cc * taps * (Cc // groups), ("TN", taps * (Cc // groups), Cc // groups, S * T, groups, True, 1, False))

    # ---- row-wise encoder kernels (csrc/rowops.hip) ----------------------------------------------
    def layernorm_fwd(self, x, R, Cc, gamma, beta, eps, y, mean, rstd):
        _chk(x, gamma, beta, y, mean, rstd)
        _lib.check(self.lib.a2f_layernorm_fwd(_ptr(x), R, Cc, _ptr(gamma), _ptr(beta), eps, _ptr(y), _ptr(mean), _ptr(rstd), self._stream(x)),
                   "a2f_layernorm_fwd")

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, R, Cc, dx, dgamma, dbeta, dx_colsum=None):
        _chk(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, dx_colsum)
        nb = self.lib.a2f_layernorm_bwd_ws_bytes(R, Cc)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.a2f_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), R, Cc, _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                              _ptr(dx_colsum), _ptr(ws), nb, self._stream(x)), "a2f_layernorm_bwd")

    def layernorm_fwd_img(self, x, R, Cc, gamma, beta, eps, y, y_img, mean, rstd):
        """layernorm_fwd that also (y given) or only (y None) writes y as a split-32 image."""
        _chk(x, gamma, beta, y, y_img, mean, rstd)
        _lib.check(self.lib.a2f_layernorm_fwd_img(_ptr(x), R, Cc, _ptr(gamma), _ptr(beta), eps, _ptr(y), _ptr(y_img), _ptr(mean), _ptr(rstd),
                                                  self._stream(x)), "a2f_layernorm_fwd_img")

    def layernorm_bwd_img(self, dy, x, mean, rstd, gamma, R, Cc, dx, dgamma, dbeta, dxm_colsum=None, keep=None, dxm_img=None):
        """layernorm_bwd + the masked gradient dxm = keep ? dx * scale : 0 (keep = (bits, scale); None: dxm = dx) as an image and its column sums."""
        _chk(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, dxm_colsum, dxm_img)
        nb = self.lib.a2f_layernorm_bwd_ws_bytes(R, Cc)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
        bits, wpr, scale = (None, 0, 1.0)
        if keep is not None:
            _chk_bits(keep[0])
            bits, wpr, scale = keep[0].data_ptr(), keep[0].shape[-1], float(keep[1])
        _lib.check(self.lib.a2f_layernorm_bwd_img(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), R, Cc, _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                                  _ptr(dxm_colsum), bits, wpr, scale, _ptr(dxm_img), _ptr(ws), nb, self._stream(x)), "a2f_layernorm_bwd_img")

    def colsum_img(self, x, R, Cc, ld, out):
        _chk(x, out)
        nb = self.lib.a2f_colsum_ws_bytes(R, Cc)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.a2f_colsum_img(_ptr(x), R, Cc, ld, _ptr(out), _ptr(ws), nb, self._stream(x)), "a2f_colsum_img")

    def colsum(self, x, R, Cc, ld, out, x_off=0):
        _chk(x, out)
        nb = self.lib.a2f_colsum_ws_bytes(R, Cc)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.a2f_colsum(_ptr(x, x_off), R, Cc, ld, _ptr(out), _ptr(ws), nb, self._stream(x)), "a2f_colsum")

    def gelu_bwd(self, dy, z, out):
        _chk(dy, z, out)
        _lib.check(self.lib.a2f_gelu_bwd(_ptr(dy), _ptr(z), dy.numel(), _ptr(out), self._stream(dy)), "a2f_gelu_bwd")

    # ---- feature-encoder layer 0 (csrc/featenc.hip) ----------------------------------------------
    def featenc0_fwd(self, wave, S, N, w, gamma, beta, Cc, eps, out, stats, out_img=False):
        """out_img: `out` receives the split-32 image of the layer output instead of fp32 rows."""
        _chk(wave, w, gamma, beta, out, stats)
        nb = self.lib.a2f_featenc0_ws_bytes(S, N, Cc)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=wave.device)
        _lib.check(self.lib.a2f_featenc0_fwd_img(_ptr(wave), S, N, _ptr(w), _ptr(gamma), _ptr(beta), Cc, eps, _ptr(out), int(out_img), _ptr(stats), _ptr(ws), nb,
                                                 self._stream(wave)), "a2f_featenc0_fwd")

    def featenc0_bwd(self, wave, S, N, w, gamma, beta, Cc, stats, dout, dw, dgamma, dbeta, dy=None):
        _chk(wave, w, gamma, beta, stats, dout, dw, dgamma, dbeta, dy)
        nb = self.lib.a2f_featenc0_ws_bytes(S, N, Cc)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=wave.device)
        _lib.check(self.lib.a2f_featenc0_bwd(_ptr(wave), S, N, _ptr(w), _ptr(gamma), _ptr(beta), Cc, _ptr(stats), _ptr(dout), _ptr(dw), _ptr(dgamma),
                                             _ptr(dbeta), _ptr(dy), _ptr(ws), nb, self._stream(wave)), "a2f_featenc0_bwd")

    # ---- MFMA flash attention (csrc/attention.hip) -----------------------------------------------
    def _attn_prec(self, precision):
        if precision is not None:
            return precision
        return self.mm_precision if self.attn_precision is None else self.attn_precision

    def attn_fwd(self, qkv, S, T, H, ld, scale, ctx, ldo, lse, precision=None):
        _chk(qkv, ctx, lse)
        ev = self._prof_begin("attn")
        _lib.check(self.lib.a2f_attn_fwd(_ptr(qkv), S, T, H, ld, scale, self._attn_prec(precision), _ptr(ctx), ldo,
                                         _ptr(lse), self._stream(qkv)), "a2f_attn_fwd")
        self._prof_end("attn", ev)

    def attn_bwd(self, qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, dqkv, precision=None):
        _chk(qkv, ctx, dctx, lse, dqkv)
        ws = torch.empty(S * H * T, dtype=torch.float32, device=qkv.device)
        ev = self._prof_begin("attn")
        _lib.check(self.lib.a2f_attn_bwd(_ptr(qkv), _ptr(ctx), _ptr(dctx), _ptr(lse), S, T, H, ld, ldo, scale,
                                         self._attn_prec(precision), _ptr(dqkv), _ptr(ws), self._stream(qkv)), "a2f_attn_bwd")
        self._prof_end("attn", ev)

    def attn_fwd_drop(self, qkv, S, T, H, ld, scale, ctx, ldo, lse, keep_q, keep_scale, precision=None):
        _chk(qkv, ctx, lse)
        _chk_bits(keep_q)
        ev = self._prof_begin("attn")
        _lib.check(self.lib.a2f_attn_fwd_drop(_ptr(qkv), S, T, H, ld, scale, self._attn_prec(precision), _ptr(ctx), ldo,
                                              _ptr(lse), keep_q.data_ptr(), float(keep_scale), self._stream(qkv)), "a2f_attn_fwd_drop")
        self._prof_end("attn", ev)

    def attn_bwd_drop(self, qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, dqkv, keep_q, keep_k, keep_scale, precision=None):
        _chk(qkv, ctx, dctx, lse, dqkv)
        _chk_bits(keep_q, keep_k)
        ws = torch.empty(S * H * T, dtype=torch.float32, device=qkv.device)
        ev = self._prof_begin("attn")
        _lib.check(self.lib.a2f_attn_bwd_drop(_ptr(qkv), _ptr(ctx), _ptr(dctx), _ptr(lse), S, T, H, ld, ldo, scale,
                                              self._attn_prec(precision), _ptr(dqkv), _ptr(ws), keep_q.data_ptr(),
                                              keep_k.data_ptr(), float(keep_scale), self._stream(qkv)), "a2f_attn_bwd_drop")
        self._prof_end("attn", ev)

    def attn_fwd_img(self, qkv, S, T, H, ld, scale, ctx, ldo, lse, ctx_img, keep=None, precision=None, qkv_img=False):
        """attention forward that also writes the context as a split-32 image; keep = (query-major bits, scale) or None.
        qkv_img: `qkv` holds the split-32 image the q|k|v projection wrote (nothing is converted in the kernel)."""
        _chk(qkv, ctx, lse, ctx_img)
        kq, ks = (None, 1.0)
        if keep is not None:
            _chk_bits(keep[0])
            kq, ks = keep[0].data_ptr(), float(keep[1])
        ev = self._prof_begin("attn")
        _lib.check(self.lib.a2f_attn_fwd_img(_ptr(qkv), int(bool(qkv_img)), S, T, H, ld, scale, self._attn_prec(precision), _ptr(ctx), ldo, _ptr(lse), kq, ks,
                                             _ptr(ctx_img), self._stream(qkv)), "a2f_attn_fwd_img")
        self._prof_end("attn", ev)

    def attn_bwd_img(self, qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, dqkv_img, keep=None, precision=None, inputs_img=False):
        """attention backward whose dqkv is written as a split-32 image; keep = (query-major bits, key-major bits, scale) or None.
        inputs_img: `qkv` and `dctx` hold split-32 images (ctx stays fp32)."""
        _chk(qkv, ctx, dctx, lse, dqkv_img)
        ws = torch.empty(S * H * T, dtype=torch.float32, device=qkv.device)
        kq, kk, ks = (None, None, 1.0)
        if keep is not None:
            _chk_bits(keep[0], keep[1])
            kq, kk, ks = keep[0].data_ptr(), keep[1].data_ptr(), float(keep[2])
        ev = self._prof_begin("attn")
        _lib.check(self.lib.a2f_attn_bwd_img(_ptr(qkv), _ptr(ctx), _ptr(dctx), int(bool(inputs_img)), _ptr(lse), S, T, H, ld, ldo, scale, self._attn_prec(precision),
                                             _ptr(dqkv_img), _ptr(ws), kq, kk, ks, 1, self._stream(qkv)), "a2f_attn_bwd_img")
        self._prof_end("attn", ev)

    # ---- dropout keep masks (csrc/dropout.hip) ---------------------------------------------------
    def attn_keep_words(self, T):
        return 2 * ((T + 63) // 64)

    def keep_bits(self, seed, site, R, Cc, p, bits):
        """seed: a Python int, or a one-element int64 HIP tensor (read at run time: graph-captured steps)."""
        _chk_bits(bits)
        if torch.is_tensor(seed):
            assert seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1
            _lib.check(self.lib.a2f_keep_bits_dev(seed.data_ptr(), int(site), R, Cc, float(p), bits.data_ptr(), bits.shape[-1], self._stream(bits)), "a2f_keep_bits_dev")
            return
        _lib.check(self.lib.a2f_keep_bits(int(seed), int(site), R, Cc, float(p), bits.data_ptr(), bits.shape[-1], self._stream(bits)), "a2f_keep_bits")

    def keep_bits_attn(self, seed, site, SH, T, p, qmajor, kmajor):
        _chk_bits(qmajor, kmajor)
        if torch.is_tensor(seed):
            assert seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1
            _lib.check(self.lib.a2f_keep_bits_attn_dev(seed.data_ptr(), int(site), SH, T, float(p), qmajor.data_ptr(), kmajor.data_ptr(), self._stream(qmajor)),
                       "a2f_keep_bits_attn_dev")
            return
        _lib.check(self.lib.a2f_keep_bits_attn(int(seed), int(site), SH, T, float(p), qmajor.data_ptr(), kmajor.data_ptr(), self._stream(qmajor)),
                   "a2f_keep_bits_attn")

    def dropout_apply(self, x, bits, R, Cc, ld, scale, y):
        _chk(x, y)
        _chk_bits(bits)
        _lib.check(self.lib.a2f_dropout_apply(_ptr(x), bits.data_ptr(), R, Cc, ld, bits.shape[-1], float(scale), _ptr(y), self._stream(x)),
                   "a2f_dropout_apply")

    # ---- a2f_wgrad_f32 ------------------------------------------------------------------------
    def wgrad_supported(self, M, N):
        return bool(self.lib.a2f_wgrad_supported(M, N))

    def wgrad_item(self, A, B, Cout, M, N, K, a, b, ldc, *, ones_out=None, mask=None, unpack_kh=0):
        """One problem of ``wgrad_batch``: C[M,N] = sum_k A(m,k) B(k,n) for tiny outputs (see a2f_hip.h); overwrites C /
        ones_out, deterministic.  ``unpack_kh``: store column kh*Ci+ci at ci*KH+kh (nn.Conv2d weight layout)."""
        _chk(A, B, Cout, ones_out)
        ws = torch.empty(self.lib.a2f_wgrad_ws_floats(M, N, K), device=Cout.device, dtype=torch.float32)
        d = GemmDesc()
        d.A, d.a_off, d.a_rs, d.a_cs = _ptr(A), a[0], a[1], a[2]
        d.B, d.b_off, d.b_rs, d.b_cs = _ptr(B), b[0], b[1], b[2]
        d.C, d.ldc = _ptr(Cout), ldc
        d.M, d.N, d.K = M, N, K
        d.ones_out = _ptr(ones_out)
        d.split_k = 1
        d.precision = self.wgrad_precision
        d.c_unpack_kh = unpack_kh
        if mask is not None:
            d.mask_operand, d.mask_period, d.mask_phase, d.mask_lo, d.mask_hi = mask
        d.splitk_ws = _ptr(ws)
        return d, (ws, Cout)

    def wgrad_batch(self, items, tag=None):
        """Run up to 8 ``wgrad_item`` problems per launch pair (a2f_wgrad_batch_f32)."""
        for i in range(0, len(items), 8):
            chunk = items[i:i + 8]
            arr = (GemmDesc * len(chunk))(*[it[0] for it in chunk])
            ev = self._prof_begin(tag)
            _lib.check(self.lib.a2f_wgrad_batch_f32(arr, len(chunk), self._stream(chunk[0][1][1])), "a2f_wgrad_batch_f32")
            self._prof_end(tag, ev)

    def wgrad(self, A, B, Cout, M, N, K, a, b, ldc, *, ones_out=None, mask=None, tag=None, unpack_kh=0):
        self.wgrad_batch([self.wgrad_item(A, B, Cout, M, N, K, a, b, ldc, ones_out=ones_out, mask=mask, unpack_kh=unpack_kh)], tag=tag)

    # ---- loss ---------------------------------------------------------------------------------
    def vocaloss_fwd(self, pred, gt, B, n, k_rec, k_vel, out3):
        _chk(pred, gt, out3)
        nb = self.lib.a2f_vocaloss_ws_bytes(B, n)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=pred.device)
        ev = self._prof_begin("loss_fwd")
        _lib.check(self.lib.a2f_vocaloss_fwd(_ptr(pred), _ptr(gt), B, n, k_rec, k_vel, _ptr(out3), _ptr(ws), nb,
                                             self._stream(pred)), "a2f_vocaloss_fwd")
        self._prof_end("loss_fwd", ev)

    def vocaloss_bwd(self, pred, gt, g3, B, n, k_rec, k_vel, grad):
        _chk(pred, gt, g3, grad)
        ev = self._prof_begin("loss_bwd")
        _lib.check(self.lib.a2f_vocaloss_bwd(_ptr(pred), _ptr(gt), _ptr(g3), B, n, k_rec, k_vel, _ptr(grad),
                                             self._stream(pred)), "a2f_vocaloss_bwd")
        self._prof_end("loss_bwd", ev)

    def head_fused(self, H, W, bias, templ, template_rows, gt, B, Bloss, n, K, k_rec, k_vel, pred, out3, dH, dW, dbias, ws=None):
        _chk(H, W, bias, templ, gt, pred, out3, dH, dW, dbias)
        nb = self.lib.a2f_head_fused_ws_bytes(B, n, K)
        if ws is None:
            ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=H.device)
        ev = self._prof_begin("head_fused")
        _lib.check(self.lib.a2f_head_fused(_ptr(H), _ptr(W), _ptr(bias), _ptr(templ), template_rows, _ptr(gt), B, Bloss, n, K,
                                           k_rec, k_vel, _ptr(pred), _ptr(out3), _ptr(dH), _ptr(dW), _ptr(dbias),
                                           _ptr(ws), nb, self._stream(H)), "a2f_head_fused")
        self._prof_end("head_fused", ev)

    def profile_head(self, on: bool):
        _lib.check(self.lib.a2f_profile_enable(int(on)), "a2f_profile_enable")

    def profile_head_ms(self):
        """(mean ms, launches) of head_fused_kernel alone since profile_head(True); call after torch.cuda.synchronize()."""
        ms, n = C.c_float(0), C.c_int32(0)
        _lib.check(self.lib.a2f_profile_head_ms(C.byref(ms), C.byref(n)), "a2f_profile_head_ms")
        return ms.value, n.value

    # ---- packing ------------------------------------------------------------------------------
    def voca_pack(self, x, one_hot, B, n_onehot, a0, cat72):
        _chk(x, one_hot, a0, cat72)
        _lib.check(self.lib.a2f_voca_pack(_ptr(x), _ptr(one_hot), B, n_onehot, _ptr(a0), _ptr(cat72), self._stream(x)), "a2f_voca_pack")

    def voca_trunk_fwd(self, x, one_hot, B, n_onehot, conv_w, conv_b, fc_w, fc_b, a0, a1, a2, a3, cat72, z1, z2, h3):
        _chk(x, one_hot, *conv_w, *conv_b, *fc_w, *fc_b, a0, a1, a2, a3, cat72, z1, z2, h3)
        arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(self.lib.a2f_voca_trunk_fwd(_ptr(x), _ptr(one_hot), B, n_onehot, arr(conv_w), arr(conv_b), arr(fc_w), arr(fc_b),
                                               _ptr(a0), _ptr(a1), _ptr(a2), _ptr(a3), _ptr(cat72), _ptr(z1), _ptr(z2), _ptr(h3),
                                               self._stream(x)), "a2f_voca_trunk_fwd")

    def voca_trunk_fwd_seq(self, feats, n_seq, seq_len, one_hot, n_onehot, conv_w, conv_b, fc_w, fc_b, a0, a1, a2, a3, cat72, z1, z2, h3):
        _chk(feats, one_hot, *conv_w, *conv_b, *fc_w, *fc_b, a0, a1, a2, a3, cat72, z1, z2, h3)
        arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _lib.check(self.lib.a2f_voca_trunk_fwd_seq(_ptr(feats), n_seq, seq_len, _ptr(one_hot), n_onehot, arr(conv_w), arr(conv_b), arr(fc_w),
                                                   arr(fc_b), _ptr(a0), _ptr(a1), _ptr(a2), _ptr(a3), _ptr(cat72), _ptr(z1), _ptr(z2),
                                                   _ptr(h3), self._stream(feats)), "a2f_voca_trunk_fwd_seq")

    def interp_linear(self, x, N, L_in, C_, L_out, y, backward=False):
        """forward: x (N,L_in,C) -> y (N,L_out,C); backward=True: x = dy (N,L_out,C) -> y = dx (N,L_in,C)."""
        _chk(x, y)
        fn = self.lib.a2f_interp_linear_bwd if backward else self.lib.a2f_interp_linear_fwd
        _lib.check(fn(_ptr(x), N, L_in, C_, L_out, _ptr(y), self._stream(x)), "a2f_interp_linear")

    def voca_trunk_bwd_data(self, dh3, z2, cat, a3, a2, a1, fw3, fw1, fw0, cw4, cw3, cw2, B, g_z2, g_z1, g_c4, g_c3, g_c2, g_c1):
        ts = (dh3, z2, cat, a3, a2, a1, fw3, fw1, fw0, cw4, cw3, cw2)
        outs = (g_z2, g_z1, g_c4, g_c3, g_c2, g_c1)
        _chk(*ts, *outs)
        _lib.check(self.lib.a2f_voca_trunk_bwd_data(*[_ptr(x) for x in ts], B, *[_ptr(x) for x in outs], self._stream(dh3)),
                   "a2f_voca_trunk_bwd_data")

    def a2m_pack(self, x, one_hot, B, n_onehot, a0, cat, cat_ld, cat_off):
        _chk(x, one_hot, a0, cat)
        _lib.check(self.lib.a2f_a2m_pack(_ptr(x), _ptr(one_hot), B, n_onehot, _ptr(a0), _ptr(cat), cat_ld, cat_off,
                                         self._stream(x)), "a2f_a2m_pack")

    def conv_weight_pack(self, src, dst, Co, Ci, KH, direction):
        _chk(src, dst)
        _lib.check(self.lib.a2f_conv_weight_pack(_ptr(src), _ptr(dst), Co, Ci, KH, direction, self._stream(src)), "a2f_conv_weight_pack")

    # ---- batch norm ---------------------------------------------------------------------------
    def bn_fwd(self, x, R, Cc, gamma, beta, rmean, rvar, momentum, eps, training, relu, y, save_mean, save_invstd):
        _chk(x, gamma, beta, rmean, rvar, y, save_mean, save_invstd)
        nb = self.lib.a2f_bn_ws_bytes(R, Cc)
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.a2f_bn_fwd(_ptr(x), R, Cc, _ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), momentum, eps,
                                       int(training), int(relu), _ptr(y), _ptr(save_mean), _ptr(save_invstd), _ptr(ws), nb,
                                       self._stream(x)), "a2f_bn_fwd")

    def bn_bwd(self, x, beta, dy, R, Cc, gamma, save_mean, save_invstd, training, relu, dx, dgamma, dbeta):
        """beta: the layer's shift - the ReLU mask of the forward is re-derived from x (the saved output is not read)."""
        _chk(x, beta, dy, gamma, save_mean, save_invstd, dx, dgamma, dbeta)
        nb = self.lib.a2f_bn_ws_bytes(R, Cc)
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.a2f_bn_bwd(_ptr(x), _ptr(beta), _ptr(dy), R, Cc, _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                                       int(training), int(relu), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), nb,
                                       self._stream(x)), "a2f_bn_bwd")

    def bn_fwd_conv0(self, a0, wg, cbias, P_out, R, Cc, gamma, beta, rmean, rvar, momentum, eps, training, relu, y, save_mean, save_invstd):
        """bn_fwd with x = the one-input-channel k = 3 / stride 2 convolution of a0 recomputed in every pass (x is never stored)."""
        _chk(a0, wg, cbias, gamma, beta, rmean, rvar, y, save_mean, save_invstd)
        nb = self.lib.a2f_bn_ws_bytes(R, Cc)
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=a0.device)
        _lib.check(self.lib.a2f_bn_fwd_conv0(_ptr(a0), _ptr(wg), _ptr(cbias), P_out, R, Cc, _ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), momentum, eps,
                                             int(training), int(relu), _ptr(y), _ptr(save_mean), _ptr(save_invstd), _ptr(ws), nb,
                                             self._stream(a0)), "a2f_bn_fwd_conv0")

    def bn_bwd_conv0(self, a0, wg, cbias, P_out, beta, dy, R, Cc, gamma, save_mean, save_invstd, training, relu, dx, dgamma, dbeta, dwg=None, dcbias=None):
        """dwg (C, 3), dcbias (C): the convolution's own parameter gradients from the same pass; dx may then be None (never written)."""
        _chk(a0, wg, cbias, beta, dy, gamma, save_mean, save_invstd, dx, dgamma, dbeta, dwg, dcbias)
        nb = self.lib.a2f_bn_ws_bytes(R, Cc)
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=a0.device)
        _lib.check(self.lib.a2f_bn_bwd_conv0(_ptr(a0), _ptr(wg), _ptr(cbias), P_out, _ptr(beta), _ptr(dy), R, Cc, _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                                             int(training), int(relu), _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(dwg), _ptr(dcbias), _ptr(ws), nb,
                                             self._stream(a0)), "a2f_bn_bwd_conv0")

    def relayout_batch(self, items):
        """items: (src, dst, Co, Ci, KH, Cop, Cip, mode[, pad_value]) - a2f_relayout_batch: packed / transposed-packed weights (modes 0 / 1, padded elements = pad_value)
        and unpacked weight gradients (mode 2) of many layers in one launch (<= 32 items per launch; longer lists are cut)."""
        for i0 in range(0, len(items), 32):
            chunk = items[i0:i0 + 32]
            arr = (_lib.RelayoutItem * len(chunk))()
            for a, it in zip(arr, chunk):
                src, dst = it[0], it[1]
                _chk(src, dst)
                a.src, a.dst = _ptr(src), _ptr(dst)
                a.Co, a.Ci, a.KH, a.Cop, a.Cip, a.mode = (int(v) for v in it[2:8])
                a.pad_value = float(it[8]) if len(it) > 8 else 0.0
                need_src = a.Co * a.Ci * a.KH if a.mode != 2 else a.Cop * a.KH * a.Cip
                need_dst = a.Cop * a.KH * a.Cip if a.mode != 2 else a.Co * a.Ci * a.KH
                if src.numel() < need_src or dst.numel() < need_dst:
                    raise A2FError(f"relayout_batch: item of {src.numel()} -> {dst.numel()} elements is smaller than its extents say ({need_src} -> {need_dst})")
            _lib.check(self.lib.a2f_relayout_batch(arr, len(chunk), self._stream(chunk[0][0])), "a2f_relayout_batch")

    # ---- FaceFormer recurrence ----------------------------------------------------------------
    def ff_save_floats(self, S, T):
        return self.lib.a2f_ff_save_bytes(S, T) // 4

    def ff_decoder_fwd(self, params, pe, cross, style, S, T, period, y, save, drop=None):
        """drop = (row_bits int32 (S*T, 10), attn_bits int32 (S*4*T, TW), keep_scale): the train-mode variant (per-row masks)."""
        _chk(params, pe, cross, style, y, save)
        if drop is not None:
            _chk_bits(drop[0], drop[1])
            _lib.check(self.lib.a2f_ff_decoder_fwd_drop(_ptr(params), _ptr(pe), _ptr(cross), _ptr(style), S, T, period, _ptr(y), _ptr(save),
                                                        save.numel() * 4, drop[0].data_ptr(), drop[1].data_ptr(), float(drop[2]), self._stream(y)),
                       "a2f_ff_decoder_fwd_drop")
            return
        _lib.check(self.lib.a2f_ff_decoder_fwd(_ptr(params), _ptr(pe), _ptr(cross), _ptr(style), S, T, period, _ptr(y), _ptr(save),
                                               save.numel() * 4, self._stream(y)), "a2f_ff_decoder_fwd")

    def ff_decoder_bwd(self, params, dy, S, T, period, save, dparams, dcross, dstyle, drop=None):
        _chk(params, dy, save, dparams, dcross, dstyle)
        if drop is not None:
            _chk_bits(drop[0], drop[1])
            _lib.check(self.lib.a2f_ff_decoder_bwd_drop(_ptr(params), _ptr(dy), S, T, period, _ptr(save), save.numel() * 4, _ptr(dparams), _ptr(dcross),
                                                        _ptr(dstyle), drop[0].data_ptr(), drop[1].data_ptr(), float(drop[2]), self._stream(dy)),
                       "a2f_ff_decoder_bwd_drop")
            return
        _lib.check(self.lib.a2f_ff_decoder_bwd(_ptr(params), _ptr(dy), S, T, period, _ptr(save), save.numel() * 4,
                                               _ptr(dparams), _ptr(dcross), _ptr(dstyle), self._stream(dy)), "a2f_ff_decoder_bwd")

    # ---- optimiser ----------------------------------------------------------------------------
    def adam_step(self, p, g, m, v, n, lr, b1, b2, eps, wd, grad_scale, step):
        _chk(p, g, m, v)
        _lib.check(self.lib.a2f_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, b1, b2, eps, wd, grad_scale, step,
                                          self._stream(p)), "a2f_adam_step")


    # ---- measurement helpers -----------------------------------------------------------------
    def stream_copy(self, dst, src):
        _chk(dst, src)
        _lib.check(self.lib.a2f_stream_copy(_ptr(dst), _ptr(src), dst.numel(), self._stream(dst)), "a2f_stream_copy")

    def stream_triad(self, dst, a, b):
        _chk(dst, a, b)
        _lib.check(self.lib.a2f_stream_triad(_ptr(dst), _ptr(a), _ptr(b), dst.numel(), self._stream(dst)), "a2f_stream_triad")


_BACKEND = None


def backend():
    """The process-wide backend; created on first use.  Raises A2FError if liba2f_hip.so is missing."""
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def set_backend(b):
    """TEST HOOK ONLY (tests/emu_backend.py injects a CPU emulator of the C ABI to check host logic)."""
    global _BACKEND
    old = _BACKEND
    _BACKEND = b
    return old


# =================================================================================================
# Layer composers (pure index arithmetic over backend.gemm)
# =================================================================================================
def split_k_for(rows: int, tiles: int = 1) -> int:
    """K-splits for a weight-gradient GEMM whose contraction runs over `rows` (frames x positions)."""
    want = max(1, min(512, (rows + 255) // 256))
    return max(1, min(want, max(1, 1024 // max(tiles, 1))))


def _splitk_ws(ref: torch.Tensor, split: int, M: int, N: int):
    """Slab workspace for deterministic split-K (None when a single split does the whole contraction)."""
    if split <= 1:
        return None
    return torch.empty(split * M * (N + 1), device=ref.device, dtype=torch.float32)


def linear_fwd(be, x, W, bias, out, M, N, K, *, ldx=None, ldc=None, act=ACT_NONE, aux=None, ldaux=0, aux_mode=AUX_NONE, tag=None):
    """out[M,N] = act(x[M,K] W[N,K]^T + bias)  (torch.nn.Linear layout)."""
    be.gemm(x, W, out, M, N, K, (0, ldx or K, 1), (0, 1, K), ldc or N, bias=bias, act=act, aux=aux, ldaux=ldaux, aux_mode=aux_mode, tag=tag)


def linear_bwd_input(be, dy, W, dx, M, N, K, *, ld_dy=None, n_cols=None, ldc=None, aux=None, ldaux=0, aux_mode=AUX_NONE):
    """dx[M,:n_cols] = (dy[M,N] W[N,K])[:, :n_cols], optionally times an activation derivative (aux)."""
    nc = n_cols or K
    be.gemm(dy, W, dx, M, nc, N, (0, ld_dy or N, 1), (0, K, 1), ldc or nc, aux=aux, ldaux=ldaux, aux_mode=aux_mode)


def linear_wgrad_item(be, dy, x, dW, db, M, N, K, *, ld_dy=None, ldx=None):
    """``wgrad_batch`` item for dW[N,K] = dy[M,N]^T x[M,K], db[N] = colsum(dy)."""
    return be.wgrad_item(dy, x, dW, N, K, M, (0, 1, ld_dy or N), (0, ldx or K, 1), K, ones_out=db)


def conv_wgrad_item(be, act_in, dout, dw, dbias, R, P_out, Cin, Co, *, KH=3, stride=2, pad=1, ld_dout=None):
    """``wgrad_batch`` item for the weight gradient of the strided implicit-GEMM convolution, stored straight into the
    nn.Conv2d layout dw (Co, Cin, KH, 1); dbias = colsum(dout)."""
    mask = (2, P_out, 0, 0, pad * Cin) if pad else None
    return be.wgrad_item(dout, act_in, dw, Co, KH * Cin, R, (0, 1, ld_dout or Co), (-pad * Cin, stride * Cin, 1), KH * Cin,
                         ones_out=dbias, mask=mask, unpack_kh=KH)


def linear_bwd_weight(be, dy, x, dW, db, M, N, K, *, ld_dy=None, ldx=None, tag=None):
    """dW[N,K] = dy[M,N]^T x[M,K];  db[N] = colsum(dy).  Overwrites dW/db (slab split-K, deterministic)."""
    if be.wgrad_supported(N, K):
        be.wgrad(dy, x, dW, N, K, M, (0, 1, ld_dy or N), (0, ldx or K, 1), K, ones_out=db, tag=tag)
        return
    tiles = ((N + 127) // 128) * ((K + 1 + 127) // 128)
    sk = split_k_for(M, tiles)
    be.gemm(dy, x, dW, N, K, M, (0, 1, ld_dy or N), (0, ldx or K, 1), K, ones_out=db, split_k=sk, tag=tag,
            splitk_ws=_splitk_ws(dW, sk, N, K))


def conv_fwd(be, act_in, wg, bias, out, R, P_out, Cin, Co, *, KH=3, stride=2, pad=1, ldc=None, c_off=0, relu=True):
    """Strided 1-D convolution over channels-last rows as an implicit GEMM.

    act_in: (groups, P_in, Cin) flattened, P_in = stride*P_out; output row r = (group, p_out), R = groups*P_out.
    The im2col row of r is the contiguous window act_in[(r*stride - pad)*Cin : +KH*Cin]; the left zero padding of the
    first position of every group is a mask on the first pad*Cin window elements.
    wg: (Co, KH*Cin) packed as k = kh*Cin + ci.
    """
    mask = (1, P_out, 0, 0, pad * Cin) if pad else None
    be.gemm(act_in, wg, out, R, Co, KH * Cin, (-pad * Cin, stride * Cin, 1), (0, 1, KH * Cin), ldc or Co, c_off=c_off,
            bias=bias, act=ACT_RELU if relu else ACT_NONE, mask=mask)


def conv_bwd_weight(be, act_in, dout, dwg, dbias, R, P_out, Cin, Co, *, KH=3, stride=2, pad=1, ld_dout=None):
    """dwg[Co, KH*Cin] = dout[R,Co]^T windows(act_in);  dbias = colsum(dout).  Outputs zeroed by the caller."""
    mask = (2, P_out, 0, 0, pad * Cin) if pad else None
    if be.wgrad_supported(Co, KH * Cin):
        be.wgrad(dout, act_in, dwg, Co, KH * Cin, R, (0, 1, ld_dout or Co), (-pad * Cin, stride * Cin, 1), KH * Cin,
                 ones_out=dbias, mask=mask)
        return
    tiles = ((Co + 127) // 128) * ((KH * Cin + 1 + 127) // 128)
    sk = split_k_for(R, tiles)
    be.gemm(dout, act_in, dwg, Co, KH * Cin, R, (0, 1, ld_dout or Co), (-pad * Cin, stride * Cin, 1), KH * Cin,
            ones_out=dbias, split_k=sk, mask=mask, splitk_ws=_splitk_ws(dwg, sk, Co, KH * Cin))


def conv_k3s2_bwd_input(be, dout, wg, d_in, R, P_out, Cin, Co, *, relu_of=None):
    """Data gradient of the k=3, stride=2, pad=1 convolution, gather form (no atomics, no col2im buffer).

    d_in: (groups, 2*P_out, Cin).  Input position 2p (even) is touched only by tap kh=1 of output p; position 2p+1 (odd)
    by tap kh=2 of output p and tap kh=0 of output p+1 (if it exists in the same group).
    relu_of: the forward activation stored at d_in's addresses; when given, the result is masked by (act > 0).
    """
    C3 = 3 * Cin
    am = AUX_RELU_BWD if relu_of is not None else AUX_NONE
    # even positions
    be.gemm(dout, wg, d_in, R, Cin, Co, (0, Co, 1), (Cin, C3, 1), 2 * Cin, aux=relu_of, ldaux=2 * Cin, aux_mode=am)
    # odd positions
    if P_out == 1:
        be.gemm(dout, wg, d_in, R, Cin, Co, (0, Co, 1), (2 * Cin, C3, 1), 2 * Cin, c_off=Cin, aux=relu_of, aux_off=Cin,
                ldaux=2 * Cin, aux_mode=am)
    else:
        be.gemm(dout, wg, d_in, R, Cin, Co, (0, Co, 1), (2 * Cin, C3, 1), 2 * Cin, c_off=Cin)
        be.gemm(dout, wg, d_in, R, Cin, Co, (Co, Co, 1), (0, C3, 1), 2 * Cin, c_off=Cin, accumulate=True,
                aux=relu_of, aux_off=Cin, ldaux=2 * Cin, aux_mode=am, mask=(1, P_out, P_out - 1, 0, Co))


def conv_nonoverlap_bwd_input(be, dout, wg, d_in, R, Cin, Co, KH, *, relu_of=None):
    """Data gradient when stride == kernel and pad == 0 (audio2face.py:45): d_in[r, kh*Cin+ci] = dout[r,:] wg[:, kh*Cin+ci]."""
    am = AUX_RELU_BWD if relu_of is not None else AUX_NONE
    be.gemm(dout, wg, d_in, R, KH * Cin, Co, (0, Co, 1), (0, KH * Cin, 1), KH * Cin, aux=relu_of, ldaux=KH * Cin, aux_mode=am)


# =================================================================================================
# Layer composers over backend.mm (bf16-MFMA GEMM family): Linear and channels-last Conv1d, forward / data / weight
# gradients.  Activations are (sequence, position, channel) rows; a strided convolution is an implicit GEMM whose A
# lines are overlapping windows of the activation buffer (no im2col), see include/a2f_hip.h (a2f_mm).
# =================================================================================================
_BIG = (1 << 31) - 16


def mm_linear_fwd(be, x, W, bias, out, M, N, K, *, act=MM_ACT_NONE, pre=None, aux=None, aux_mode=MM_AUX_NONE, tag="gemm_mfma"):
    """out[M,N] = act(x[M,K] W[N,K]^T + bias) (+ aux);  pre (optional) receives the pre-activation values."""
    be.mm(MMOp(x, K), MMOp(W, K), out, M, N, K, N, bias=bias, act=act, C2=pre, aux=aux, ldaux=N, aux_mode=aux_mode, tag=tag)


def mm_linear_dgrad(be, dy, W, dx, M, N, K, *, aux=None, aux_mode=MM_AUX_NONE, tag="gemm_mfma", keep=None, WT=None):
    """dx[M,K] = dy[M,N] W[N,K], optionally times an activation derivative evaluated on aux[M,K].  The weight is read through a
    transposed copy (K, N) so that BOTH operands are k-contiguous: the row-image staging path (one ds_read_b128 per fragment) is
    12-15 % faster than the transposing one, and the copy is 1/M of the product's traffic."""
    if W.is_cuda and M >= 1024:                                # WT: the caller's own (K, N) copy of this step's weight
        be.mm(MMOp(dy, N), MMOp(WT if WT is not None else W.t().contiguous(), N), dx, M, K, N, K, aux=aux, ldaux=K, aux_mode=aux_mode, tag=tag, keep=keep)
        return
    be.mm(MMOp(dy, N), MMOp(W, K, trans=True), dx, M, K, N, K, aux=aux, ldaux=K, aux_mode=aux_mode, tag=tag, keep=keep)


def mm_linear_wgrad(be, dy, x, dW, M, N, K, *, tag="gemm_mfma", db=None):
    """dW[N,K] = dy[M,N]^T x[M,K]: the contraction runs over the M rows of the batch (split-K slabs, fixed-order fold).  db (N), optional:
    the bias gradient = column sums of dy, collected inside the same kernel."""
    be.mm(MMOp(dy, N, trans=True), MMOp(x, K, trans=True), dW, N, K, M, K, split_k=0, tag=tag, a_rowsum=db)


def mm_conv_fwd(be, x, Wg, y, S, Lin, Lout, Ci, Co, KH, stride, *, pad=0, bias=None, act=MM_ACT_NONE, pre=None, tag="gemm_mfma", precision=None):
    """Conv1d over channels-last rows: x (S,Lin,Ci) -> y (S,Lout,Co), Wg (Co, KH*Ci) with k = kh*Ci + ci; `pad` zero
    positions on the left of every sequence."""
    A = MMOp(x, stride * Ci, lines_per_seq=Lout, seq_stride=Lin * Ci, off=-pad * Ci, seq_elems=Lin * Ci)
    be.mm(A, MMOp(Wg, KH * Ci), y, S * Lout, Co, KH * Ci, Co, bias=bias, act=act, C2=pre, tag=tag, precision=precision)


def mm_conv_wgrad(be, x, dy, dWg, S, Lin, Lout, Ci, Co, KH, stride, *, pad=0, tag="gemm_mfma", db=None):
    """dWg[Co, KH*Ci] = sum over (sequence, position) of dy[row, co] * window(x)[row, :].  db (Co), optional: the bias gradient =
    column sums of dy, collected inside the same kernel."""
    A = MMOp(dy, Co, trans=True)
    B = MMOp(x, stride * Ci, trans=True, lines_per_seq=Lout, seq_stride=Lin * Ci, off=-pad * Ci, seq_elems=Lin * Ci)
    be.mm(A, B, dWg, Co, KH * Ci, S * Lout, KH * Ci, split_k=0, tag=tag, a_rowsum=db)


def mm_conv_dgrad(be, dy, Wg, dx, S, Lin, Lout, Ci, Co, KH, stride, *, pad=0, tag="gemm_mfma", mul=None, WgT=None):
    """dx (S,Lin,Ci) from dy (S,Lout,Co), gather form (no atomics, no col2im buffer); positions no window touches must be
    zero in dx on entry.  Supported: KH == stride, pad = 0 (non-overlapping windows) and KH = 3, stride = 2 with pad = 0
    (Lin >= 2 Lout + 1) or pad = 1 (Lin = 2 Lout).  mul (S,Lin,Ci), optional: every written element is multiplied by it in the
    epilogue (the saved activation derivative of the layer below: dx then IS that layer's pre-activation gradient)."""
    ex = dict(aux=mul, aux_mode=MM_AUX_MUL_LIKE_C) if mul is not None else {}
    # plain products read the packed weight through a transposed copy (KH*Ci, Co): both operands k-contiguous (see mm_linear_dgrad)
    if not (Wg.is_cuda and S * Lout >= 1024):
        WgT = None
    elif WgT is None:                                         # (a caller that keeps its own (KH*Ci, Co) copy of this step's weight passes it)
        WgT = Wg.t().contiguous()
    if KH == stride and pad == 0:
        Bop = MMOp(WgT, Co) if WgT is not None else MMOp(Wg, KH * Ci, trans=True)
        be.mm(MMOp(dy, Co), Bop, dx, S * Lout, KH * Ci, Co, KH * Ci, c_rows_per_seq=Lout, c_seq_stride=Lin * Ci, tag=tag, **ex)
        return
    if KH == 3 and stride == 2 and pad in (0, 1):
        # positions touched by two taps (pad 0: even j = 2q from dy[q-1] (kh=2), dy[q] (kh=0); pad 1: odd j = 2q+1 from dy[q] (kh=2),
        # dy[q+1] (kh=0)): a two-row window of dy against [W_kh=2 ; W_kh=0]
        assert Lin >= 2 * Lout + 1 if pad == 0 else Lin == 2 * Lout
        rows = Lout + 1 if pad == 0 else Lout
        A = MMOp(dy, Co, lines_per_seq=rows, seq_stride=Lout * Co, off=-Co if pad == 0 else 0, seq_elems=Lout * Co)
        B = MMOp(Wg, 3 * Ci, trans=True, lines_per_seq=Co, seq_stride=-2 * Ci, off=2 * Ci, seq_elems=_BIG)
        be.mm(A, B, dx, S * rows, Ci, 2 * Co, 2 * Ci, c_off=0 if pad == 0 else Ci, c_rows_per_seq=rows, c_seq_stride=Lin * Ci, tag=tag, **ex)
        # positions touched by the middle tap only: dy[q] against W_kh=1
        Bmid = MMOp(WgT, Co, ptr_off=Ci * Co) if WgT is not None else MMOp(Wg, 3 * Ci, trans=True, ptr_off=Ci)
        be.mm(MMOp(dy, Co), Bmid, dx, S * Lout, Ci, Co, 2 * Ci, c_off=Ci if pad == 0 else 0, c_rows_per_seq=Lout,
              c_seq_stride=Lin * Ci, tag=tag, **ex)
        return
    raise A2FError(f"mm_conv_dgrad: unsupported kernel/stride/pad {KH}/{stride}/{pad}")
