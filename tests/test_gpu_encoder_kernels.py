"""GPU: the row-wise encoder kernels (csrc/rowops.hip) and the MFMA flash attention (csrc/attention.hip) against the float64
emulator of their contracts (include/a2f_hip.h), through the C ABI."""
import pytest
import torch

from a2f_amd import ops
from emu_backend import EmuBackend
from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("R,C", [(1, 64), (7, 512), (600, 768), (9601, 768), (130, 1024), (33, 4)])
def test_layernorm_fwd_bwd(R, C):
    x, dy = rnd(R, C, seed=1, scale=2.0) + 0.3, rnd(R, C, seed=2)
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    res = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        new = lambda *s: torch.zeros(*s, device=dev)
        y, mean, rstd, dx, dg, db, dc = new(R, C), new(R), new(R), new(R, C), new(C), new(C), new(C)
        be.layernorm_fwd(x.to(dev), R, C, gamma.to(dev), beta.to(dev), 1e-5, y, mean, rstd)
        be.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, gamma.to(dev), R, C, dx, dg, db, dc)
        res.append([t.cpu() for t in (y, mean, rstd, dx, dg, db, dc)])
    for g, r, name in zip(res[0], res[1], ("y", "mean", "rstd", "dx", "dgamma", "dbeta", "dx_colsum")):
        if name == "dx_colsum":                       # sums to ~0 by construction (dx is mean-free per row only, not per column): absolute scale
            assert (g - r).abs().max() < 1e-4 * (res[1][3].abs().sum(0).max() + 1e-6), name
        else:
            assert rel_l2(g, r) < 2e-5, name


def test_colsum_and_gelu_bwd():
    R, C, ld = 9600, 768, 2304
    x = rnd(R, ld, seed=1)
    dy, z = rnd(R, C, seed=2), rnd(R, C, seed=3, scale=2.0)
    res = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        out, out2 = torch.zeros(C, device=dev), torch.zeros(R, C, device=dev)
        be.colsum(x.to(dev), R, C, ld, out, x_off=768)
        be.gelu_bwd(dy.to(dev), z.to(dev), out2)
        res.append((out.cpu(), out2.cpu()))
    assert (res[0][0] - res[1][0]).abs().max() < 1e-3 * res[1][0].abs().max() + 1e-3
    assert rel_l2(res[0][1], res[1][1]) < 1e-5


@pytest.mark.parametrize("S,T,H", [(1, 30, 12), (2, 64, 2), (1, 600, 12), (3, 129, 4), (1, 1000, 1)])
@pytest.mark.parametrize("precision,tol", [(1, 3e-5), (0, 2e-2)])
def test_attention_fwd_bwd(S, T, H, precision, tol):
    ld, ldo = 3 * 64 * H, 64 * H
    qkv = rnd(S * T, ld, seed=1, scale=1.0)
    qkv[:, :64 * H] *= 2.0                                       # sharper softmax than unit-variance scores
    dctx = rnd(S * T, ldo, seed=2)
    res = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        ctx, lse, dqkv = torch.zeros(S * T, ldo, device=dev), torch.zeros(S, H, T, device=dev), torch.zeros(S * T, ld, device=dev)
        be.attn_fwd(qkv.to(dev), S, T, H, ld, 0.125, ctx, ldo, lse, precision=precision)
        be.attn_bwd(qkv.to(dev), ctx, dctx.to(dev), lse, S, T, H, ld, ldo, 0.125, dqkv, precision=precision)
        res.append([t.cpu() for t in (ctx, lse, dqkv[:, :64 * H], dqkv[:, 64 * H:128 * H], dqkv[:, 128 * H:])])
    for g, r, name in zip(res[0], res[1], ("ctx", "lse", "dq", "dk", "dv")):
        assert rel_l2(g, r) < tol, name


def test_attention_forced_max_jump():
    """Online-softmax rescale path: one key dominates late in the sequence (the running maximum jumps by ~40 at a chosen tile)."""
    S, T, H = 1, 300, 1
    qkv = rnd(T, 192, seed=5, scale=0.5)
    qkv[:, 0:64] = 1.0                    # all queries equal
    qkv[200, 64:128] = 40.0               # key 200 dominates: scaled score 0.125 * 64 * 40 = 320 ... bounded by scale below
    qkv[200, 64:128] *= 0.125
    res = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        ctx, lse = torch.zeros(T, 64, device=dev), torch.zeros(1, 1, T, device=dev)
        be.attn_fwd(qkv.to(dev), S, T, H, 192, 0.125, ctx, 64, lse, precision=1)
        res.append((ctx.cpu(), lse.cpu()))
    assert rel_l2(res[0][0], res[1][0]) < 3e-5 and rel_l2(res[0][1], res[1][1]) < 1e-5


@pytest.mark.parametrize("S,N", [(1, 110), (2, 1283), (3, 8013), (2, 160008)])
def test_featenc0_fwd_bwd(S, N):
    """conv1d(1->512, k=10, s=5) + GroupNorm(512 groups) + GELU, channels-last, against torch's float64 ops (emulator)."""
    C = 512
    L = (N - 10) // 5 + 1
    wave, w = rnd(S, N, seed=1), rnd(C, 1, 10, seed=2, scale=0.3)
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    dout = rnd(S, L, C, seed=5)
    res = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        new = lambda *s: torch.zeros(*s, device=dev)
        out, stats, dw, dg, db, dy = new(S, L, C), new(S, 2, C), new(C, 1, 10), new(C), new(C), new(S, L, C)
        be.featenc0_fwd(wave.to(dev), S, N, w.to(dev), gamma.to(dev), beta.to(dev), C, 1e-5, out, stats)
        be.featenc0_bwd(wave.to(dev), S, N, w.to(dev), gamma.to(dev), beta.to(dev), C, stats, dout.to(dev), dw, dg, db, dy)
        res.append([t.cpu() for t in (out, stats[:, 0], dw, dg, db, dy)])
    for g, r, name in zip(res[0], res[1], ("out", "mean", "dw", "dgamma", "dbeta", "dy")):
        assert rel_l2(g, r) < 2e-5, name


# ---------------------------------------------------------------------------------------------------------------------------
# train-mode dropout: the keep-mask generator, the streaming apply pass, the a2f_mm epilogue and the attention kernels
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,C,p", [(1, 4, 0.1), (77, 768, 0.1), (600, 3072, 0.25), (5, 100, 0.0)])
def test_keep_bits_is_philox_bit_for_bit(R, C, p):
    """a2f_keep_bits against the numpy restatement of its contract (Philox4x32-10, counter = (column / 8, row, site), keep iff
    the element's 16-bit half word >= round(p 2^16)): every word equal; the bits beyond C are zero."""
    from emu_backend import philox_keep_words
    wpr = (C + 31) // 32 + (1 if R == 77 else 0)                # a padded row pitch is allowed
    seed, site = 0x1234_5678_9ABC_DEF0 >> 1, 9
    bits = torch.full((R, wpr), -1, dtype=torch.int32, device=DEV)
    ops.HipBackend().keep_bits(seed, site, R, C, p, bits)
    assert torch.equal(bits.cpu(), torch.from_numpy(philox_keep_words(seed, site, R, C, p, wpr)))


@pytest.mark.parametrize("SH,T", [(3, 30), (2, 64), (24, 600), (1, 129)])
def test_keep_bits_attn_two_layouts_describe_one_mask(SH, T):
    from emu_backend import philox_keep_words, unpack_keep_bits
    be = ops.HipBackend()
    TW = be.attn_keep_words(T)
    q = torch.zeros(SH * T, TW, dtype=torch.int32, device=DEV)
    k = torch.zeros(SH, TW, 32 * TW, dtype=torch.int32, device=DEV)
    be.keep_bits_attn(77, 6, SH, T, 0.1, q, k)
    assert torch.equal(q.cpu(), torch.from_numpy(philox_keep_words(77, 6, SH * T, T, 0.1, TW)))
    mq = unpack_keep_bits(q.cpu(), T).view(SH, T, T)                                                       # [sh][query][key]
    mk = unpack_keep_bits(k.cpu().transpose(1, 2).reshape(SH * 32 * TW, TW), 32 * TW).view(SH, 32 * TW, 32 * TW)     # [sh][key][query]
    assert torch.equal(mk[:, :T, :T].transpose(1, 2), mq)
    assert not bool(mk[:, T:].any()) and not bool(mk[:, :, T:].any())
    assert abs(float(mq.float().mean()) - 0.9) < 0.02


def test_dropout_apply_and_mm_epilogue_keep():
    from a2f_amd.dropout import pack_keep_bits
    from a2f_amd.ops import MM_ACT_GELU, MM_AUX_ADD, MMOp
    M, N, K = 333, 200, 96
    gen = torch.Generator().manual_seed(3)
    mask = torch.rand(M, N, generator=gen) > 0.2
    x, W, b, res = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.2), rnd(N, seed=6), rnd(M, N, seed=7)
    outs = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        bits = pack_keep_bits(mask).to(dev)
        y = torch.zeros(M, N, device=dev)
        be.dropout_apply(res.to(dev), bits, M, N, N, 1.25, y)
        c, c2 = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
        be.mm(MMOp(x.to(dev), K), MMOp(W.to(dev), K), c, M, N, K, N, bias=b.to(dev), act=MM_ACT_GELU, C2=c2, aux=res.to(dev), ldaux=N, aux_mode=MM_AUX_ADD,
              keep=(bits, 1.25))
        cs = torch.zeros(N, K, device=dev)                     # the same through the split-K fold kernel's epilogue
        be.mm(MMOp(res.to(dev), N, trans=True), MMOp(x.to(dev), K, trans=True), cs, N, K, M, K, split_k=3, keep=(pack_keep_bits(mask[:N, :K].contiguous()).to(dev), 2.0))
        outs.append([t.cpu() for t in (y, c, c2, cs)])
    assert torch.equal(outs[0][0], torch.where(mask, res * 1.25, torch.zeros(())))
    for a, e, name in zip(outs[0][1:], outs[1][1:], ("C", "C2", "split-K C")):
        assert rel_l2(a, e) < 3e-5, name
    assert bool((outs[0][1] == res)[~mask].all())              # dropped elements: exactly the residual


@pytest.mark.parametrize("S,T,H", [(1, 30, 12), (2, 64, 2), (1, 600, 12), (3, 129, 4)])
@pytest.mark.parametrize("precision,tol", [(1, 3e-5), (0, 2e-2)])
def test_attention_with_probability_dropout(S, T, H, precision, tol):
    """a2f_attn_fwd_drop / a2f_attn_bwd_drop against softmax -> mask / (1 - p) -> @ v and its autograd in float64, with an
    injected (torch-drawn) mask packed in both layouts."""
    from a2f_amd.dropout import pack_keep_bits_attn
    ld, ldo = 3 * 64 * H, 64 * H
    qkv = rnd(S * T, ld, seed=1, scale=1.0)
    qkv[:, :64 * H] *= 2.0
    dctx = rnd(S * T, ldo, seed=2)
    mask = torch.rand(S * H, T, T, generator=torch.Generator().manual_seed(8)) >= 0.1
    kq, kk = pack_keep_bits_attn(mask)
    res = []
    for be, dev in ((ops.HipBackend(), DEV), (EmuBackend(), "cpu")):
        ctx, lse, dqkv = torch.zeros(S * T, ldo, device=dev), torch.zeros(S, H, T, device=dev), torch.zeros(S * T, ld, device=dev)
        be.attn_fwd_drop(qkv.to(dev), S, T, H, ld, 0.125, ctx, ldo, lse, kq.to(dev), 1 / 0.9, precision=precision)
        be.attn_bwd_drop(qkv.to(dev), ctx, dctx.to(dev), lse, S, T, H, ld, ldo, 0.125, dqkv, kq.to(dev), kk.to(dev), 1 / 0.9, precision=precision)
        res.append([t.cpu() for t in (ctx, lse, dqkv)])
    for a, e, name in zip(res[0], res[1], ("ctx", "lse", "dqkv")):
        assert rel_l2(a, e) < tol, name
    for off, name in ((0, "dq"), (64 * H, "dk"), (128 * H, "dv")):
        assert rel_l2(res[0][2][:, off:off + 64 * H], res[1][2][:, off:off + 64 * H]) < tol, name


@pytest.mark.parametrize("S,T,H", [(1, 600, 12), (1, 3000, 12), (3, 129, 4)])
def test_attention_fp8_forward_mode(S, T, H):
    """precision 2: q k^T and P v on v_mfma_f32_32x32x16_fp8_fp8 (e4m3 operands narrowed in registers, P pre-scaled by 2^8), the mode
    BASELINE configs[4] names.  Measured, not assumed: the context error sits at ~5e-2 (e4m3 keeps 3 mantissa bits), 15x the bf16 mode
    and 50x configs[4]'s own 1e-3 bound - fp8 is selectable, bf16x3 stays the parity setting.  The backward of this mode
    runs the bf16 kernels."""
    ld, ldo = 3 * 64 * H, 64 * H
    qkv = rnd(S * T, ld, seed=1)
    dctx = rnd(S * T, ldo, seed=2)
    emu, hip = EmuBackend(), ops.HipBackend()
    ctx_e, lse_e = torch.zeros(S * T, ldo), torch.zeros(S, H, T)
    emu.attn_fwd(qkv, S, T, H, ld, 0.125, ctx_e, ldo, lse_e)
    errs = {}
    for prec in (2, 0):
        ctx, lse = torch.zeros(S * T, ldo, device=DEV), torch.zeros(S, H, T, device=DEV)
        hip.attn_fwd(qkv.to(DEV), S, T, H, ld, 0.125, ctx, ldo, lse, precision=prec)
        errs[prec] = rel_l2(ctx.cpu(), ctx_e)
        assert rel_l2(lse.cpu(), lse_e) < (3e-2 if prec == 2 else 2e-3)
    assert 1e-2 < errs[2] < 8e-2 and errs[0] < 6e-3 and errs[2] > 5 * errs[0]
    dq2, dq0 = torch.zeros(S * T, ld, device=DEV), torch.zeros(S * T, ld, device=DEV)
    ctx0, lse0 = torch.zeros(S * T, ldo, device=DEV), torch.zeros(S, H, T, device=DEV)
    hip.attn_fwd(qkv.to(DEV), S, T, H, ld, 0.125, ctx0, ldo, lse0, precision=0)
    hip.attn_bwd(qkv.to(DEV), ctx0, dctx.to(DEV), lse0, S, T, H, ld, ldo, 0.125, dq2, precision=2)
    hip.attn_bwd(qkv.to(DEV), ctx0, dctx.to(DEV), lse0, S, T, H, ld, ldo, 0.125, dq0, precision=0)
    assert torch.equal(dq2, dq0)




@pytest.mark.parametrize("S,T", [(2, 17), (3, 50), (1, 600), (2, 700), (1, 385), (9, 600), (2, 3000)])
def test_positional_convolution_kernels_against_torch_conv1d_float64(S, T):
    """csrc/posconv.hip through encoder_ops.PosConvFn (forward, data gradient, weight gradient, bias gradient) against torch's own grouped conv1d in float64
    (HF Wav2Vec2PositionalConvEmbedding: Conv1d(768, 768, 128, padding 64, groups 16), last frame dropped, GELU, + input): one block per sequence, two blocks
    (T = 700, 385: the block edge inside a sequence), sequences shorter than the kernel, the bench step's blocking (9 x 600: two blocks of 304 frames, wave rows of
    4 / 5 / 5 / 5 tiles) and the long configuration's (2 x 3000: eight blocks of 384 frames, six tiles per wave row; one or few sequences get 64-frame blocks so
    that the launch still fills the chip); and the kernels alone against the float64 emulator of their contract."""
    import torch.nn.functional as F
    from a2f_amd import encoder_ops
    C, G, K = 768, 16, 128
    h = rnd(S, T, C, seed=1)
    w = rnd(C, C // G, K, seed=2, scale=0.02)
    b = rnd(C, seed=3, scale=0.1)
    dy = rnd(S, T, C, seed=4)
    hd, wd, bd = (t.double().requires_grad_(True) for t in (h, w, b))
    ref = hd + F.gelu(F.conv1d(hd.transpose(1, 2), wd, bd, padding=K // 2, groups=G)[:, :, :-1].transpose(1, 2))
    ref.backward(dy.double())
    hg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (h, w, b))
    out = encoder_ops.PosConvFn.apply(hg, wg, bg, G)
    out.backward(dy.to(DEV))
    assert rel_l2(out.detach().cpu(), ref.detach()) < 2e-5
    for name, got, want in (("dh", hg.grad, hd.grad), ("dw", wg.grad, wd.grad), ("db", bg.grad, bd.grad)):
        assert rel_l2(got.cpu(), want) < 3e-5, name
    if S * T > 2000:
        return
    # the two entry points against the emulator (other offsets / epilogues than PosConvFn uses)
    hip, emu = ops.HipBackend(), EmuBackend()
    w2 = w.view(G, C // G, C // G, K).permute(0, 1, 3, 2).contiguous().view(C, K * (C // G))
    res = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        wi = torch.empty(C, K * (C // G), device=dev)
        be.split_img(w2.to(dev), C, K * (C // G), K * (C // G), wi, K * (C // G))
        y, z, dw = torch.zeros(S, T, C, device=dev), torch.zeros(S, T, C, device=dev), torch.zeros(G, K, C // G, C // G, device=dev)
        be.posconv_nt(h.to(dev), wi, y, S, T, C, G, K, -5, z=z, act=0)
        be.posconv_wgrad(h.to(dev), dy.to(dev), dw, S, T, C, G, K, 3)
        res.append((y.cpu(), z.cpu(), dw.cpu()))
    for g_, r_, name in zip(res[0], res[1], ("y", "z", "dw")):
        assert rel_l2(g_, r_) < 2e-5, name
