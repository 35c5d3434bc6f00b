"""CPU emulator of the liba2f_hip.so C ABI (TEST INFRASTRUCTURE ONLY).

Implements the *semantics* documented in include/a2f_hip.h with plain torch CPU ops so that
 (a) the host-side composers in ``ops.py`` / the drop-in modules can be checked against the oracle
     without a GPU (index arithmetic of the implicit-GEMM convolutions, gradient plumbing), and
 (b) the GPU tests can compare every HIP entry point against the same contract on random operands.
The product never imports this file.
"""
from __future__ import annotations

import math

import torch


def _flat(t):
    return t.reshape(-1)


def _window(t, off, rs, cs, R, Ccols):
    """Matrix view M[r,c] = flat[off + r*rs + c*cs] with out-of-range elements read as 0."""
    flat = _flat(t).double()
    r = torch.arange(R).view(-1, 1)
    c = torch.arange(Ccols).view(1, -1)
    idx = off + r * rs + c * cs
    ok = (idx >= 0) & (idx < flat.numel())
    return torch.where(ok, flat[idx.clamp(0, flat.numel() - 1)], torch.zeros((), dtype=torch.double)), ok


def pack_keep_bits(mask):
    """bool (R, C) -> int32 (R, ceil(C/32)) words, bit b of word w <-> column 32 w + b."""
    R, Cc = mask.shape
    wpr = (Cc + 31) // 32
    m = torch.zeros(R, wpr * 32, dtype=torch.int64)
    m[:, :Cc] = mask.to(torch.int64)
    w = (m.reshape(R, wpr, 32) << torch.arange(32, dtype=torch.int64)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def unpack_keep_bits(bits, Cc):
    """int32 (R, wpr) words -> bool (R, Cc)."""
    w = bits.to(torch.int64) & 0xFFFFFFFF
    return (((w.unsqueeze(-1) >> torch.arange(32, dtype=torch.int64)) & 1) != 0).reshape(bits.shape[0], -1)[:, :Cc]


def philox_keep_words(seed, site, R, Cc, p, wpr):
    """numpy restatement of a2f_keep_bits (include/a2f_hip.h): Philox4x32-10, key = seed, counter = (column / 8, row, site, 0); the four output words
    are eight 16-bit randoms (element 8 g + 2 e: low half of word e, 8 g + 2 e + 1: high half); an element keeps iff its random >= round(p * 2^16).
    Returns int32 (R, wpr)."""
    import numpy as np
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask32 = np.uint64(0xFFFFFFFF)
    ng = wpr * 4
    c0 = np.broadcast_to(np.arange(ng, dtype=np.uint64)[None, :], (R, ng)).copy()
    c1 = np.broadcast_to(np.arange(R, dtype=np.uint64)[:, None], (R, ng)).copy()
    c2 = np.full((R, ng), site, dtype=np.uint64)
    c3 = np.zeros((R, ng), dtype=np.uint64)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask32, p1 >> np.uint64(32), p1 & mask32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & mask32, (k1 + W1) & mask32
    t = float(p) * 65536.0 + 0.5
    thresh = np.uint64(0 if t <= 0 else min(int(t), 65535))
    words4 = np.stack([c0, c1, c2, c3], axis=-1)                              # (R, ng, 4)
    halves = np.stack([words4 & np.uint64(0xFFFF), words4 >> np.uint64(16)], axis=-1)   # (R, ng, 4, 2): element 8 g + 2 e + half
    keep = (halves >= thresh).reshape(R, ng * 8)
    keep[:, Cc:] = False
    words = (keep.reshape(R, wpr, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    return words.view(np.int32)


def img_encode(x):
    """fp32 (..., C) with C % 32 == 0 -> float32 tensor of the same shape whose BYTES are the split-32 image (include/a2f_hip.h, a2f_mm_operand.format 1):
    per 32 elements 32 bf16 hi = RNE(x) then 32 bf16 lo = RNE(x - hi)."""
    x = x.float()
    assert x.shape[-1] % 32 == 0
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    blk = torch.stack((hi.reshape(*x.shape[:-1], -1, 32), lo.reshape(*x.shape[:-1], -1, 32)), dim=-2)       # (..., C/32, 2, 32) bf16
    return blk.contiguous().view(torch.float32).reshape(x.shape)


def img_decode(img):
    """the float64 values hi + lo an image stands for."""
    blk = img.contiguous().view(torch.bfloat16).reshape(*img.shape[:-1], -1, 2, 32)
    return (blk[..., 0, :].double() + blk[..., 1, :].double()).reshape(img.shape)


class EmuBackend:
    name = "emu"

    def __init__(self, strict_bounds: bool = True):
        self.gemm_precision = 0          # accepted and ignored: the emulator is the float64 contract
        self.strict = strict_bounds
        self.calls = []

    # ------------------------------------------------------------------ gemm
    def gemm(self, A, B, Cout, M, N, K, a, b, ldc, *, c_off=0, bias=None, act=0, aux=None, aux_off=0, ldaux=0, aux_mode=0,
             accumulate=False, split_k=1, ones_out=None, mask=None, tag=None, splitk_ws=None):
        self.calls.append(("gemm", M, N, K))
        Am, Aok = _window(A, a[0], a[1], a[2], M, K)
        Bm, Bok = _window(B, b[0], b[1], b[2], K, N)
        if mask is not None:
            op, period, phase, lo, hi = mask
            if op == 1:
                rows = (torch.arange(M) % period == phase).view(-1, 1)
                cols = ((torch.arange(K) >= lo) & (torch.arange(K) < hi)).view(1, -1)
                m = rows & cols
                Am = Am.masked_fill(m, 0.0); Aok = Aok | m
            elif op == 2:
                rows = (torch.arange(K) % period == phase).view(-1, 1)
                cols = ((torch.arange(N) >= lo) & (torch.arange(N) < hi)).view(1, -1)
                m = rows & cols
                Bm = Bm.masked_fill(m, 0.0); Bok = Bok | m
        if self.strict:      # every unmasked element must be inside its buffer (a kernel would fault otherwise)
            assert bool(Aok.all()), "gemm emu: A window reads outside its buffer"
            assert bool(Bok.all()), "gemm emu: B window reads outside its buffer"
        v = Am @ Bm
        cflat = _flat(Cout)
        r = torch.arange(M).view(-1, 1)
        cidx = c_off + r * ldc + torch.arange(N).view(1, -1)
        assert int(cidx.max()) < cflat.numel(), "gemm emu: C write outside its buffer"
        if split_k > 1:
            assert act == 0 and aux_mode == 0 and bias is None and not accumulate
            if splitk_ws is not None:                                  # slab mode: deterministic overwrite
                assert splitk_ws.numel() >= split_k * M * (N + 1)
                cflat[cidx] = v.float()
                if ones_out is not None:
                    ones_out[:M] = Am.sum(dim=1).float()
                return
            cflat[cidx] = (cflat[cidx].double() + v).float()          # atomics into zeroed C
            if ones_out is not None:
                ones_out[:M] += Am.sum(dim=1).float()
            return
        if bias is not None:
            v = v + bias.double().view(1, -1)
        if accumulate:
            v = v + cflat[cidx].double()
        if act == 1:
            v = v.clamp_min(0)
        elif act == 2:
            v = torch.tanh(v)
        if aux_mode:
            ax = _flat(aux)[aux_off + r * ldaux + torch.arange(N).view(1, -1)].double()
            if aux_mode == 1:
                v = torch.where(ax > 0, v, torch.zeros_like(v))
            elif aux_mode == 2:
                v = v * (1 - ax * ax)
            else:
                v = v + ax
        cflat[cidx] = v.float()
        if ones_out is not None:
            s = Am.sum(dim=1).float()
            ones_out[:M] = ones_out[:M] + s if accumulate else s

    # ------------------------------------------------------------------ a2f_mm (bf16-MFMA GEMM family), float64 contract
    mm_precision = 1
    attn_precision = None

    @staticmethod
    def _mm_matrix(op, nlines, ncontig):
        """(nlines, ncontig) float64 matrix of an a2f_mm_operand by the header's contract (zeros outside the windows)."""
        if getattr(op, "img", False):            # image operands: rows of ld elements, ptr_off a multiple of 32
            assert op.ptr_off % 32 == 0 and op.ld % 32 == 0
            flat = img_decode(op.t.reshape(-1)[op.ptr_off:].reshape(-1, 32)).reshape(-1)
        else:
            flat = op.t.reshape(-1)[op.ptr_off:].double()
        i = torch.arange(nlines).view(-1, 1)
        j = torch.arange(ncontig).view(1, -1)
        if op.lines_per_seq > 0:
            s, q = i // op.lines_per_seq, i % op.lines_per_seq
            e = q * op.ld + j + op.off
            ok = (e >= 0) & (e < op.seq_elems)
            idx = s * op.seq_stride + e
        else:
            idx = i * op.ld + j
            ok = torch.ones_like(idx, dtype=torch.bool)
        ok = ok & (idx >= 0) & (idx < flat.numel())
        return torch.where(ok, flat[idx.clamp(0, flat.numel() - 1)], torch.zeros((), dtype=torch.double))

    def mm(self, A, B, Cout, M, N, K, ldc, *, c_off=0, c_rows_per_seq=0, c_seq_stride=0, C2=None, bias=None, aux=None, ldaux=0, act=0,
           aux_mode=0, precision=None, split_k=1, batch=1, a_bstride=0, b_bstride=0, c_bstride=0, bias_bstride=0, aux_bstride=0,
           accumulate=False, alpha=1.0, tag=None, keep=None, a_rowsum=None, c_img=False, c_colsum=None):
        import copy
        if c_img:                                      # image output: whole 32-blocks; the values land as hi / lo pairs
            assert getattr(A, "img", False) and ldc % 32 == 0 and c_off % 32 == 0 and c_seq_stride % 32 == 0 and N % 32 == 0 and batch == 1 and not accumulate
            tmp = torch.zeros(Cout.numel())
            self.mm(A, B, tmp, M, N, K, ldc, c_off=c_off, c_rows_per_seq=c_rows_per_seq, c_seq_stride=c_seq_stride, C2=C2, bias=bias, aux=aux, ldaux=ldaux, act=act,
                    aux_mode=aux_mode, precision=precision, alpha=alpha, keep=keep, c_colsum=c_colsum)
            enc = img_encode(tmp.view(-1, 32))
            mm_ = torch.arange(M).view(-1, 1)
            crow_ = mm_ * ldc if c_rows_per_seq <= 0 else (mm_ // c_rows_per_seq) * c_seq_stride + (mm_ % c_rows_per_seq) * ldc
            blocks = ((c_off + crow_ + torch.arange(0, N, 32).view(1, -1)) // 32).reshape(-1)
            Cout.reshape(-1, 32)[blocks] = enc[blocks]
            return
        self.calls.append(("mm", M, N, K, batch))
        cflat = Cout.reshape(-1)
        c2flat = C2.reshape(-1) if C2 is not None else None
        m = torch.arange(M).view(-1, 1)
        crow = m * ldc if c_rows_per_seq <= 0 else (m // c_rows_per_seq) * c_seq_stride + (m % c_rows_per_seq) * ldc
        for bz in range(batch):
            Ab, Bb = copy.copy(A), copy.copy(B)
            Ab.ptr_off, Bb.ptr_off = A.ptr_off + bz * a_bstride, B.ptr_off + bz * b_bstride
            Am = self._mm_matrix(Ab, K, M).t() if A.trans else self._mm_matrix(Ab, M, K)
            Bm = self._mm_matrix(Bb, K, N) if B.trans else self._mm_matrix(Bb, N, K).t()
            v = alpha * (Am @ Bm)
            if a_rowsum is not None:
                assert batch == 1 and A.trans and B.trans
                a_rowsum.copy_(Am.sum(dim=1).float())
            if bias is not None:
                v = v + bias.reshape(-1)[bz * bias_bstride: bz * bias_bstride + N].double().view(1, -1)
            cidx = c_off + bz * c_bstride + crow + torch.arange(N).view(1, -1)
            assert int(cidx.max()) < cflat.numel(), "mm emu: C write outside its buffer"
            if act == 4:                                   # GELU, C2 = gelu'(v)
                cdf = 0.5 * (1 + torch.erf(v / math.sqrt(2.0)))
                if c2flat is not None:
                    c2flat[cidx] = (cdf + v * torch.exp(-0.5 * v * v) / math.sqrt(2 * math.pi)).float()
                v = v * cdf
            else:
                if c2flat is not None:
                    c2flat[cidx] = v.float()
                if act == 1:
                    v = v.clamp_min(0)
                elif act == 3:
                    v = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
            if keep is not None:
                assert batch == 1
                v = torch.where(unpack_keep_bits(keep[0], N)[:M], v * float(keep[1]), torch.zeros_like(v))
            if aux_mode == 4:                              # aux laid out like C (c_off and the row map included)
                v = v * aux.reshape(-1)[cidx].double()
            elif aux_mode:
                ax = aux.reshape(-1)[bz * aux_bstride + m * ldaux + torch.arange(N).view(1, -1)].double()
                if aux_mode == 1:
                    v = v + ax
                elif aux_mode == 2:
                    v = v * (0.5 * (1 + torch.erf(ax / math.sqrt(2.0))) + ax * torch.exp(-0.5 * ax * ax) / math.sqrt(2 * math.pi))
                else:
                    v = torch.where(ax > 0, v, torch.zeros_like(v))
            if accumulate:
                v = v + cflat[cidx].double()
            cflat[cidx] = v.float()
            if c_colsum is not None:
                assert batch == 1
                c_colsum.copy_(v.sum(0).float())

    def split_img(self, x, R, Cc, ld, out, ldo, transpose=False):
        self.calls.append(("split_img", R, Cc, bool(transpose)))
        src = x.reshape(-1)[: (R - 1) * ld + Cc]
        m = src[torch.arange(R).view(-1, 1) * ld + torch.arange(Cc).view(1, -1)]
        if transpose:
            m = m.t().contiguous()
        rows, cols = m.shape
        pad = torch.zeros(rows, ((cols + 31) // 32) * 32)
        pad[:, :cols] = m
        enc = img_encode(pad)
        of = out.reshape(-1)
        of[(torch.arange(rows).view(-1, 1) * ldo + torch.arange(pad.shape[1]).view(1, -1)).reshape(-1)] = enc.reshape(-1)

    # ------------------------------------------------------------------ row-wise encoder kernels / attention (float64 contracts)
    def layernorm_fwd(self, x, R, Cc, gamma, beta, eps, y, mean, rstd):
        xd = x.reshape(R, Cc).double()
        mu = xd.mean(dim=1, keepdim=True)
        var = ((xd - mu) ** 2).mean(dim=1, keepdim=True)
        rs = 1.0 / torch.sqrt(var + eps)
        y.reshape(R, Cc).copy_(((xd - mu) * rs * gamma.double() + beta.double()).float())
        mean.copy_(mu.reshape(-1).float()); rstd.copy_(rs.reshape(-1).float())

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, R, Cc, dx, dgamma, dbeta, dx_colsum=None):
        xd, dyd = x.reshape(R, Cc).double(), dy.reshape(R, Cc).double()
        xh = (xd - mean.double().view(-1, 1)) * rstd.double().view(-1, 1)
        g = dyd * gamma.double()
        d = rstd.double().view(-1, 1) * (g - g.mean(dim=1, keepdim=True) - xh * (g * xh).mean(dim=1, keepdim=True))
        dx.reshape(R, Cc).copy_(d.float())
        dgamma.copy_((dyd * xh).sum(0).float()); dbeta.copy_(dyd.sum(0).float())
        if dx_colsum is not None:
            dx_colsum.copy_(d.sum(0).float())

    def layernorm_fwd_img(self, x, R, Cc, gamma, beta, eps, y, y_img, mean, rstd):
        tmp = torch.empty(R, Cc)
        self.layernorm_fwd(x, R, Cc, gamma, beta, eps, tmp, mean, rstd)
        if y is not None:
            y.reshape(R, Cc).copy_(tmp)
        if y_img is not None:
            y_img.reshape(R, Cc).copy_(img_encode(tmp))

    def layernorm_bwd_img(self, dy, x, mean, rstd, gamma, R, Cc, dx, dgamma, dbeta, dxm_colsum=None, keep=None, dxm_img=None):
        self.layernorm_bwd(dy, x, mean, rstd, gamma, R, Cc, dx, dgamma, dbeta, None)
        d = dx.reshape(R, Cc).clone()
        if keep is not None:
            d = torch.where(unpack_keep_bits(keep[0], Cc)[:R], d * float(keep[1]), torch.zeros_like(d))
        if dxm_colsum is not None:
            dxm_colsum.copy_(d.double().sum(0).float())
        if dxm_img is not None:
            dxm_img.reshape(R, Cc).copy_(img_encode(d))

    def colsum_img(self, x, R, Cc, ld, out):
        v = img_decode(x.reshape(-1)[: R * ld].reshape(R, ld))[:, :Cc]
        out.copy_(v.sum(0).float())

    def colsum(self, x, R, Cc, ld, out, x_off=0):
        flat = x.reshape(-1)[x_off:].double()
        idx = torch.arange(R).view(-1, 1) * ld + torch.arange(Cc).view(1, -1)
        out.copy_(flat[idx].sum(0).float())

    def interp_linear(self, x, N, L_in, C_, L_out, y, backward=False):
        """a2f_interp_linear_fwd / _bwd: align_corners=True linear interpolation along the row axis of (N, L, C) and its adjoint."""
        import torch.nn.functional as F
        if not backward:
            y.copy_(F.interpolate(x.reshape(N, L_in, C_).double().transpose(1, 2), size=L_out, align_corners=True, mode="linear").transpose(1, 2).float())
            return
        with torch.enable_grad():
            z = torch.zeros(N, L_in, C_, dtype=torch.float64, requires_grad=True)
            F.interpolate(z.transpose(1, 2), size=L_out, align_corners=True, mode="linear").transpose(1, 2).backward(x.reshape(N, L_out, C_).double())
        y.copy_(z.grad.float())

    def posconv_nt(self, x, w_img, y, S, T, Cc, groups, taps, off, *, bias=None, aux=None, z=None, act=0, tag=None):
        """a2f_posconv_nt in float64: grouped window product with the decoded weight image (C, taps * C/groups), [g*Cg+co][kk*Cg+ci]."""
        self.calls.append(("posconv_nt", S, T, off))
        Cg = Cc // groups
        W = img_decode(w_img.reshape(Cc, taps * Cg)).double().view(groups, Cg, taps, Cg)                  # [g][co][kk][ci]
        xd = x.reshape(S, T, groups, Cg).double()
        xp = torch.zeros(S, T + 2 * taps, groups, Cg, dtype=torch.float64)
        xp[:, taps:taps + T] = xd
        v = torch.zeros(S, T, groups, Cg, dtype=torch.float64)
        for kk in range(taps):
            v += torch.einsum("stgi,goi->stgo", xp[:, taps + off + kk: taps + off + kk + T], W[:, :, kk, :])
        v = v.reshape(S, T, Cc)
        if bias is not None:
            v = v + bias.double().view(1, 1, Cc)
        if z is not None:
            z.copy_(v.float().view_as(z))
        if act:
            v = 0.5 * v * (1 + torch.erf(v / math.sqrt(2.0)))
        if aux is not None:
            v = v + aux.reshape(S, T, Cc).double()
        y.copy_(v.float().view_as(y))

    def posconv_wgrad(self, x, dz, dw, S, T, Cc, groups, taps, pad, tag=None):
        self.calls.append(("posconv_wgrad", S, T, pad))
        Cg = Cc // groups
        xp = torch.zeros(S, T + 2 * taps, groups, Cg, dtype=torch.float64)
        xp[:, taps:taps + T] = x.reshape(S, T, groups, Cg).double()
        dzd = dz.reshape(S, T, groups, Cg).double()
        out = torch.zeros(groups, taps, Cg, Cg, dtype=torch.float64)
        for kk in range(taps):
            out[:, kk] = torch.einsum("stgi,stgo->gio", xp[:, taps - pad + kk: taps - pad + kk + T], dzd)
        dw.copy_(out.float().view_as(dw))

    def gelu_bwd(self, dy, z, out):
        zd = z.double()
        g = 0.5 * (1 + torch.erf(zd / math.sqrt(2.0))) + zd * torch.exp(-0.5 * zd * zd) / math.sqrt(2 * math.pi)
        out.copy_((dy.double() * g).float())

    def featenc0_fwd(self, wave, S, N, w, gamma, beta, Cc, eps, out, stats, out_img=False):
        import torch.nn.functional as F
        y = F.conv1d(wave.reshape(S, 1, N).double(), w.reshape(Cc, 1, 10).double(), stride=5)          # (S,C,L)
        mu = y.mean(dim=2, keepdim=True)
        var = ((y - mu) ** 2).mean(dim=2, keepdim=True)
        rs = 1.0 / torch.sqrt(var + eps)
        u = (y - mu) * rs * gamma.double().view(1, -1, 1) + beta.double().view(1, -1, 1)
        o = F.gelu(u).transpose(1, 2).float()
        out.copy_(img_encode(o) if out_img else o)
        stats.reshape(S, 2, Cc)[:, 0].copy_(mu.squeeze(2).float()); stats.reshape(S, 2, Cc)[:, 1].copy_(rs.squeeze(2).float())

    def featenc0_bwd(self, wave, S, N, w, gamma, beta, Cc, stats, dout, dw, dgamma, dbeta, dy=None):
        import torch.nn.functional as F
        with torch.enable_grad():
            wd = w.detach().reshape(Cc, 1, 10).double().requires_grad_(True)
            gd, bd = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
            y = F.conv1d(wave.detach().reshape(S, 1, N).double(), wd, stride=5)
            y.retain_grad()
            o = F.gelu(F.group_norm(y, Cc, gd, bd, 1e-5)).transpose(1, 2)
            o.backward(dout.detach().double())
        dw.copy_(wd.grad.reshape(dw.shape).float()); dgamma.copy_(gd.grad.float()); dbeta.copy_(bd.grad.float())
        if dy is not None:
            dy.copy_(y.grad.transpose(1, 2).float())

    @staticmethod
    def _attn_split(qkv, S, T, H, ld):
        m = qkv.reshape(S, T, ld).double()
        take = lambda o: m[:, :, o:o + 64 * H].reshape(S, T, H, 64).permute(0, 2, 1, 3)      # (S,H,T,64)
        return take(0), take(64 * H), take(128 * H)

    def attn_fwd(self, qkv, S, T, H, ld, scale, ctx, ldo, lse, precision=None):
        q, k, v = self._attn_split(qkv, S, T, H, ld)
        s = torch.matmul(q * scale, k.transpose(-1, -2))
        lse.reshape(S, H, T).copy_(torch.logsumexp(s, dim=-1).float())
        o = torch.matmul(torch.softmax(s, dim=-1), v)                                        # (S,H,T,64)
        ctx.reshape(S * T, ldo)[:, :64 * H].copy_(o.permute(0, 2, 1, 3).reshape(S * T, 64 * H).float())

    def attn_bwd(self, qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, dqkv, precision=None):
        with torch.enable_grad():                         # called from inside autograd.Function.backward (grad mode off there)
            q, k, v = (t.clone().requires_grad_(True) for t in self._attn_split(qkv.detach(), S, T, H, ld))
            o = torch.matmul(torch.softmax(torch.matmul(q * scale, k.transpose(-1, -2)), dim=-1), v)
            do = dctx.detach().reshape(S, T, ldo)[:, :, :64 * H].double().reshape(S, T, H, 64).permute(0, 2, 1, 3)
            o.backward(do)
        out = dqkv.reshape(S, T, ld)
        for t, off in ((q, 0), (k, 64 * H), (v, 128 * H)):
            out[:, :, off:off + 64 * H] = t.grad.permute(0, 2, 1, 3).reshape(S, T, 64 * H).float()

    def attn_fwd_drop(self, qkv, S, T, H, ld, scale, ctx, ldo, lse, keep_q, keep_scale, precision=None):
        q, k, v = self._attn_split(qkv, S, T, H, ld)
        s = torch.matmul(q * scale, k.transpose(-1, -2))
        lse.reshape(S, H, T).copy_(torch.logsumexp(s, dim=-1).float())
        d = unpack_keep_bits(keep_q.reshape(S * H * T, -1), T).reshape(S, H, T, T).double() * float(keep_scale)
        o = torch.matmul(torch.softmax(s, dim=-1) * d, v)
        ctx.reshape(S * T, ldo)[:, :64 * H].copy_(o.permute(0, 2, 1, 3).reshape(S * T, 64 * H).float())

    def attn_bwd_drop(self, qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, dqkv, keep_q, keep_k, keep_scale, precision=None):
        TW = keep_q.reshape(S * H * T, -1).shape[1]
        dq_ = unpack_keep_bits(keep_q.reshape(S * H * T, -1), T).reshape(S, H, T, T)
        # the key-major copy must describe the same mask: word (qw, key), bit b <-> query 32 qw + b
        dk_ = unpack_keep_bits(keep_k.reshape(S * H, TW, 32 * TW).permute(0, 2, 1).reshape(S * H * 32 * TW, TW), T).reshape(S * H, 32 * TW, T)[:, :T]
        assert torch.equal(dq_.reshape(S * H, T, T), dk_.transpose(1, 2)), "attn_bwd_drop emu: the two keep-bit layouts disagree"
        with torch.enable_grad():
            q, k, v = (t.clone().requires_grad_(True) for t in self._attn_split(qkv.detach(), S, T, H, ld))
            pr = torch.softmax(torch.matmul(q * scale, k.transpose(-1, -2)), dim=-1)
            o = torch.matmul(pr * (dq_.double() * float(keep_scale)), v)
            do = dctx.detach().reshape(S, T, ldo)[:, :, :64 * H].double().reshape(S, T, H, 64).permute(0, 2, 1, 3)
            o.backward(do)
        out = dqkv.reshape(S, T, ld)
        for t, off in ((q, 0), (k, 64 * H), (v, 128 * H)):
            out[:, :, off:off + 64 * H] = t.grad.permute(0, 2, 1, 3).reshape(S, T, 64 * H).float()

    # ------------------------------------------------------------------ dropout keep masks (contract of csrc/dropout.hip)
    def attn_fwd_img(self, qkv, S, T, H, ld, scale, ctx, ldo, lse, ctx_img, keep=None, precision=None, qkv_img=False):
        if qkv_img:
            qkv = img_decode(qkv.reshape(S * T, ld))
        if keep is None:
            self.attn_fwd(qkv, S, T, H, ld, scale, ctx, ldo, lse, precision)
        else:
            self.attn_fwd_drop(qkv, S, T, H, ld, scale, ctx, ldo, lse, keep[0], keep[1], precision)
        if ctx_img is not None:
            ctx_img.reshape(S * T, ldo).copy_(img_encode(ctx.reshape(S * T, ldo)))

    def attn_bwd_img(self, qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, dqkv_img, keep=None, precision=None, inputs_img=False):
        if inputs_img:
            qkv, dctx = img_decode(qkv.reshape(S * T, ld)), img_decode(dctx.reshape(S * T, ldo))
        tmp = torch.zeros(S * T, ld)
        if keep is None:
            self.attn_bwd(qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, tmp, precision)
        else:
            self.attn_bwd_drop(qkv, ctx, dctx, lse, S, T, H, ld, ldo, scale, tmp, keep[0], keep[1], keep[2], precision)
        dqkv_img.reshape(S * T, ld).copy_(img_encode(tmp))

    def attn_keep_words(self, T):
        return 2 * ((T + 63) // 64)

    def keep_bits(self, seed, site, R, Cc, p, bits):
        seed = int(seed.item()) & 0xFFFFFFFFFFFFFFFF if torch.is_tensor(seed) else seed          # device-resident seed of the graph-captured step
        bits.copy_(torch.from_numpy(philox_keep_words(seed, site, R, Cc, p, bits.shape[-1])))

    def keep_bits_attn(self, seed, site, SH, T, p, qmajor, kmajor):
        seed = int(seed.item()) & 0xFFFFFFFFFFFFFFFF if torch.is_tensor(seed) else seed
        TW = self.attn_keep_words(T)
        words = philox_keep_words(seed, site, SH * T, T, p, TW)
        qmajor.reshape(SH * T, TW).copy_(torch.from_numpy(words))
        keep = unpack_keep_bits(torch.from_numpy(words), T).reshape(SH, T, T)                      # [sh][q][k]
        full = torch.zeros(SH, 32 * TW, 32 * TW, dtype=torch.bool)
        full[:, :T, :T] = keep
        kmajor.reshape(SH, TW, 32 * TW).copy_(pack_keep_bits(full.transpose(1, 2).reshape(SH * 32 * TW, 32 * TW)).reshape(SH, 32 * TW, TW).transpose(1, 2))

    def dropout_apply(self, x, bits, R, Cc, ld, scale, y):
        k = unpack_keep_bits(bits.reshape(R, -1), Cc)
        xv = x.reshape(-1)[: R * ld].reshape(R, ld)[:, :Cc]
        y.reshape(-1)[: R * ld].reshape(R, ld)[:, :Cc] = torch.where(k, xv * float(scale), torch.zeros_like(xv))

    def wgrad_supported(self, M, N):
        mt, nt = (M + 31) // 32, (N + 1 + 31) // 32
        return (nt <= 7) if mt in (1, 2) else (nt <= 4) if mt == 3 else (nt <= 3) if mt == 4 else False

    def wgrad(self, A, B, Cout, M, N, K, a, b, ldc, *, ones_out=None, mask=None, tag=None, unpack_kh=0):
        assert a[1] == 1 and b[2] == 1 and self.wgrad_supported(M, N)
        ws = torch.empty(2 * M * (N + 1))
        if unpack_kh:
            assert ldc == N and N % unpack_kh == 0
            tmp = torch.empty(M, N)
            self.gemm(A, B, tmp, M, N, K, a, b, N, ones_out=ones_out, mask=mask, split_k=2, splitk_ws=ws)
            ci = N // unpack_kh                                     # column kh*Ci+ci -> ci*KH+kh
            Cout.reshape(M, ci, unpack_kh).copy_(tmp.reshape(M, unpack_kh, ci).transpose(1, 2))
        else:
            self.gemm(A, B, Cout, M, N, K, a, b, ldc, ones_out=ones_out, mask=mask, split_k=2, splitk_ws=ws)

    def wgrad_item(self, *args, **kw):
        return (args, kw), None

    def wgrad_batch(self, items, tag=None):
        assert 1 <= len(items)
        for (args, kw), _ in items:
            self.wgrad(*args, **kw)

    # ------------------------------------------------------------------ loss
    def vocaloss_fwd(self, pred, gt, B, n, k_rec, k_vel, out3):
        d = (pred.reshape(B, n) - gt.reshape(B, n)).double()
        V = n // 3
        rec = (d ** 2).sum() / (B * V)
        dv = d[1::2] - d[0::2]
        vel = (dv ** 2).sum() / ((B // 2) * V)
        out3[0] = float(rec * k_rec + vel * k_vel); out3[1] = float(rec); out3[2] = float(vel)

    def vocaloss_bwd(self, pred, gt, g3, B, n, k_rec, k_vel, grad):
        d = (pred.reshape(B, n) - gt.reshape(B, n)).double()
        V = n // 3
        cr = (float(g3[0]) * k_rec + float(g3[1])) * 2.0 / (B * V)
        cv = (float(g3[0]) * k_vel + float(g3[2])) * 2.0 / ((B // 2) * V)
        dv = cv * (d[1::2] - d[0::2])
        g = cr * d
        g[0::2] -= dv
        g[1::2] += dv
        grad.reshape(B, n).copy_(g.float())

    def head_fused(self, H, W, bias, templ, template_rows, gt, B, Bloss, n, K, k_rec, k_vel, pred, out3, dH, dW, dbias):
        Hd, Wd = H.reshape(B, K).double(), W.reshape(n, K).double()
        t = templ.reshape(template_rows, n).double()
        assert B % template_rows == 0
        p = Hd @ Wd.t() + bias.double().view(1, -1) + t.repeat_interleave(B // template_rows, dim=0)
        if pred is not None:                       # NULL pred: the loss-only training variant
            pred.reshape(B, n).copy_(p.float())
        V = n // 3
        d = (p - gt.reshape(B, n).double())[:Bloss]
        rec = (d ** 2).sum() / (Bloss * V)
        dvv = d[1::2] - d[0::2]
        vel = (dvv ** 2).sum() / ((Bloss // 2) * V)
        out3[0] = float(rec * k_rec + vel * k_vel); out3[1] = float(rec); out3[2] = float(vel)
        cr = k_rec * 2.0 / (Bloss * V)
        cv = k_vel * 2.0 / ((Bloss // 2) * V)
        G = torch.zeros(B, n, dtype=torch.double)
        G[:Bloss] = cr * d
        G[0:Bloss:2] -= cv * dvv
        G[1:Bloss:2] += cv * dvv
        dH.reshape(B, K).copy_((G @ Wd).float())
        dW.reshape(n, K).copy_((G.t() @ Hd).float())
        dbias.copy_(G.sum(dim=0).float())

    # ------------------------------------------------------------------ packing
    def voca_pack(self, x, one_hot, B, n_onehot, a0, cat72):
        a0v = a0.reshape(B, 16, 37)
        a0v[:, :, :29] = x.reshape(B, 29, 16).transpose(1, 2)
        h = torch.arange(16)
        a0v[:, :, 29:] = one_hot.reshape(B, n_onehot)[:, h % 8].unsqueeze(2).expand(B, 16, 8)
        cat72.reshape(B, 72)[:, 64:] = one_hot.reshape(B, n_onehot)[:, :8]

    def a2m_pack(self, x, one_hot, B, n_onehot, a0, cat, cat_ld, cat_off):
        rows = 52 + n_onehot
        a0v = a0.reshape(B, rows, 32)
        a0v[:, :52] = x.reshape(B, 52, 32)
        r = torch.arange(n_onehot).view(-1, 1)
        c = torch.arange(32).view(1, -1)
        a0v[:, 52:] = one_hot.reshape(B, n_onehot)[:, (32 * r + c) % n_onehot]
        cat.reshape(B, cat_ld)[:, cat_off:cat_off + n_onehot] = one_hot.reshape(B, n_onehot)

    def conv_k3s2_c1_fwd(self, x, wg, bias, R, P_out, Co, z):
        xf = x.reshape(-1)[: 2 * R].double().view(R // P_out, 2 * P_out)
        xp = torch.nn.functional.pad(xf, (1, 0))                                   # left zero padding of every group
        win = torch.stack([xp[:, 0:2 * P_out:2], xp[:, 1:2 * P_out + 1:2], xp[:, 2:2 * P_out + 2:2]], -1)   # (groups, P_out, 3): x[2p-1], x[2p], x[2p+1]
        z.copy_((win.reshape(R, 3) @ wg.double().view(Co, 3).t() + bias.double().view(1, Co)).to(z.dtype).view_as(z))

    def conv_weight_pack(self, src, dst, Co, Ci, KH, direction):
        if direction == 0:
            dst.reshape(Co, KH, Ci).copy_(src.reshape(Co, Ci, KH).permute(0, 2, 1))
        else:
            dst.reshape(Co, Ci, KH).copy_(src.reshape(Co, KH, Ci).permute(0, 2, 1))

    # ------------------------------------------------------------------ batch norm (channels-last rows)
    def bn_fwd(self, x, R, Cc, gamma, beta, rmean, rvar, momentum, eps, training, relu, y, save_mean, save_invstd):
        xv = x.reshape(R, Cc).double()
        if training:
            mean = xv.mean(dim=0)
            var = xv.var(dim=0, unbiased=False)
            rmean.mul_(1 - momentum).add_((momentum * mean).float())
            unb = var * (R / max(R - 1, 1))
            rvar.mul_(1 - momentum).add_((momentum * unb).float())
        else:
            mean, var = rmean.double(), rvar.double()
        invstd = 1.0 / torch.sqrt(var + eps)
        out = (xv - mean) * invstd * gamma.double() + beta.double()
        if relu:
            out = out.clamp_min(0)
        y.reshape(R, Cc).copy_(out.float())
        save_mean.copy_(mean.float()); save_invstd.copy_(invstd.float())

    def bn_bwd(self, x, beta, dy, R, Cc, gamma, save_mean, save_invstd, training, relu, dx, dgamma, dbeta):
        xv, dyv = x.reshape(R, Cc).double(), dy.reshape(R, Cc).double()
        mean, invstd, g = save_mean.double(), save_invstd.double(), gamma.double()
        xhat = (xv - mean) * invstd
        if relu:                                           # the mask is re-derived from x: forward output > 0
            dyv = torch.where(xhat * g + beta.double() > 0, dyv, torch.zeros_like(dyv))
        db = dyv.sum(dim=0)
        dg = (dyv * xhat).sum(dim=0)
        if training:
            dxv = (g * invstd / R) * (R * dyv - db - xhat * dg)
        else:
            dxv = dyv * g * invstd
        dx.reshape(R, Cc).copy_(dxv.float()); dgamma.copy_(dg.float()); dbeta.copy_(db.float())

    def relayout_batch(self, items):
        for it in items:
            src, dst, Co, Ci, KH, Cop, Cip, mode = it[:8]
            pad = float(it[8]) if len(it) > 8 else 0.0
            if mode == 2:
                dst.reshape(-1)[: Co * Ci * KH].copy_(src.reshape(-1)[: Cop * KH * Cip].view(Cop, KH, Cip)[:Co, :, :Ci].permute(0, 2, 1).reshape(-1))
                continue
            wg = torch.full((Cop, KH, Cip), pad, dtype=src.dtype)
            wg[:Co, :, :Ci] = src.reshape(-1)[: Co * Ci * KH].view(Co, Ci, KH).permute(0, 2, 1)
            wg = wg.view(Cop, KH * Cip)
            dst.reshape(-1)[: Cop * KH * Cip].copy_((wg if mode == 0 else wg.t()).reshape(-1))

    def _conv0(self, a0, wg, cbias, P_out, R, Cc):
        z = torch.empty(R, Cc, dtype=torch.float64)
        self.conv_k3s2_c1_fwd(a0.double(), wg.double(), cbias.double(), R, P_out, Cc, z)
        return z

    def bn_fwd_conv0(self, a0, wg, cbias, P_out, R, Cc, gamma, beta, rmean, rvar, momentum, eps, training, relu, y, save_mean, save_invstd):
        self.bn_fwd(self._conv0(a0, wg, cbias, P_out, R, Cc), R, Cc, gamma, beta, rmean, rvar, momentum, eps, training, relu, y, save_mean, save_invstd)

    def bn_bwd_conv0(self, a0, wg, cbias, P_out, beta, dy, R, Cc, gamma, save_mean, save_invstd, training, relu, dx, dgamma, dbeta, dwg=None, dcbias=None):
        dxv = dx if dx is not None else torch.empty(R, Cc)
        self.bn_bwd(self._conv0(a0, wg, cbias, P_out, R, Cc), beta, dy, R, Cc, gamma, save_mean, save_invstd, training, relu, dxv, dgamma, dbeta)
        if dwg is not None:                                # the convolution's own gradients: windows a0[2r - 1 + kh] (left pad at the first row of a group)
            a = a0.double().reshape(-1, 2 * P_out)
            win = torch.stack([torch.cat([torch.zeros_like(a[:, :1]), a[:, 1:-1:2]], dim=1), a[:, 0::2], a[:, 1::2]], dim=2).reshape(R, 3)
            d = dxv.reshape(R, Cc).double()
            dwg.view(Cc, 3).copy_((d.t() @ win).float()); dcbias.copy_(d.sum(dim=0).float())

    # ------------------------------------------------------------------ FaceFormer recurrence (autograd on the spec)
    @staticmethod
    def _ff_unpack(p):
        o, D, F_ = 0, 64, 128
        def take(n, shape):
            nonlocal o
            t = p[o:o + n].view(*shape); o += n
            return t
        Wqkv = take(192 * D, (192, D)); bqkv = take(192, (192,))
        Wo = take(D * D, (D, D)); bo = take(D, (D,))
        W1 = take(F_ * D, (F_, D)); b1 = take(F_, (F_,))
        W2 = take(D * F_, (D, F_)); b2 = take(D, (D,))
        ln = take(6 * D, (6, D))
        A = take(D * D, (D, D)); c = take(D, (D,))
        return Wqkv, bqkv, Wo, bo, W1, b1, W2, b2, ln, A, c

    @classmethod
    def ff_spec(cls, params, pe, cross, style, period, drop=None):
        """Differentiable torch statement of the recurrence the HIP kernel implements (double precision).  drop = (row keep (S*T, 320)
        bool: columns 0..63 PPE, 64..127 dropout1, 128..255 activation, 256..319 dropout3; attn keep (S*4*T, T) bool; scale)."""
        import torch.nn.functional as F
        Wqkv, bqkv, Wo, bo, W1, b1, W2, b2, ln, A, c = cls._ff_unpack(params)
        S, T, D = cross.shape
        H, hd = 4, 16
        slopes = torch.tensor([2.0 ** (-2 * (h + 1)) for h in range(H)], dtype=params.dtype)
        ys = []
        for s in range(S):
            ks, vs, rows = [], [], []
            emb = style[s]
            for i in range(T):
                x = emb + pe[i % period]
                if drop is not None:
                    rk, ak, dsc = drop
                    rb = rk[s * T + i].to(params.dtype) * dsc
                    x = x * rb[:64]
                qkv = F.linear(x, Wqkv, bqkv)
                q = qkv[:D].view(H, hd) * 0.25
                ks.append(qkv[D:2 * D]); vs.append(qkv[2 * D:])
                Kc = torch.stack(ks).view(i + 1, H, hd); Vc = torch.stack(vs).view(i + 1, H, hd)
                sc = torch.einsum("hd,jhd->hj", q, Kc)
                steps = torch.div(i - torch.arange(i + 1), period, rounding_mode="floor").to(params.dtype)
                sc = sc - slopes.view(-1, 1) * steps.view(1, -1)
                a = torch.softmax(sc, dim=-1)
                if drop is not None:
                    a = a * torch.stack([ak[(s * H + h) * T + i, : i + 1] for h in range(H)]).to(params.dtype) * dsc
                ctx = torch.einsum("hj,jhd->hd", a, Vc).reshape(D)
                sa = F.linear(ctx, Wo, bo)
                y = F.layer_norm(x + (sa * rb[64:128] if drop is not None else sa), (D,), ln[0], ln[1], 1e-5)
                y = F.layer_norm(y + cross[s, i], (D,), ln[2], ln[3], 1e-5)
                hid = F.relu(F.linear(y, W1, b1))
                ff = F.linear(hid * rb[128:256] if drop is not None else hid, W2, b2)
                y = F.layer_norm(y + (ff * rb[256:320] if drop is not None else ff), (D,), ln[4], ln[5], 1e-5)
                rows.append(y)
                emb = A @ y + c + style[s]
            ys.append(torch.stack(rows))
        return torch.stack(ys)

    def ff_save_floats(self, S, T):
        return S * T * 64 * 2 + 8          # the emulator keeps (cross, style-free) inputs it needs for its autograd replay

    @staticmethod
    def _ff_drop(drop, S, T):
        if drop is None:
            return None
        return (unpack_keep_bits(drop[0].reshape(S * T, -1), 320), unpack_keep_bits(drop[1].reshape(S * 4 * T, -1), T), float(drop[2]))

    def ff_decoder_fwd(self, params, pe, cross, style, S, T, period, y, save, drop=None):
        with torch.no_grad():
            out = self.ff_spec(params.double(), pe.double(), cross.reshape(S, T, 64).double(), style.reshape(S, 64).double(), period,
                               self._ff_drop(drop, S, T))
        y.reshape(S, T, 64).copy_(out.float())
        self._ff_ctx = (pe.clone(), cross.reshape(S, T, 64).clone(), style.reshape(S, 64).clone())

    def ff_decoder_bwd(self, params, dy, S, T, period, save, dparams, dcross, dstyle, drop=None):
        pe, cross, style = self._ff_ctx
        with torch.enable_grad():                       # we are called from inside an autograd backward
            p = params.detach().double().requires_grad_(True)
            cr = cross.detach().double().requires_grad_(True)
            st = style.detach().double().requires_grad_(True)
            out = self.ff_spec(p, pe.double(), cr, st, period, self._ff_drop(drop, S, T))
            out.backward(dy.reshape(S, T, 64).double())
        dparams.copy_(p.grad.float()); dcross.reshape(S, T, 64).copy_(cr.grad.float()); dstyle.reshape(S, 64).copy_(st.grad.float())

    # ------------------------------------------------------------------ Adam
    def adam_step(self, p, g, m, v, n, lr, b1, b2, eps, wd, grad_scale, step):
        gg = g[:n] * grad_scale + wd * p[:n]
        m[:n].mul_(b1).add_(gg, alpha=1 - b1)
        v[:n].mul_(b2).addcmul_(gg, gg, value=1 - b2)
        c1 = 1 - b1 ** step
        c2 = 1 - b2 ** step
        p[:n].addcdiv_(m[:n], (v[:n].sqrt() / math.sqrt(c2)).add_(eps), value=-lr / c1)
