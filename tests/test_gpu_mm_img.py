"""GPU: the image-operand kernels of a2f_mm (csrc/gemm_img.hip: split-32 images, 256 x 256 tiles, LDS-DMA staging, ping-pong wave groups)
and a2f_split_img against the float64 emulator of the contract in include/a2f_hip.h.  The emulator decodes the images (hi + lo exactly), so
what is left on the GPU side is the dropped lo x lo term and fp32 accumulation: <= 3e-5 relative L2, the bf16x3 bar of tests/test_gpu_mm.py."""
import pytest
import torch

import a2f_amd
from a2f_amd import ops
from a2f_amd.ops import MMOp
from emu_backend import EmuBackend, img_decode, img_encode, pack_keep_bits
from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def both(fn):
    hip, emu = ops.HipBackend(), EmuBackend()
    a = [t.cpu() for t in fn(hip, DEV)]
    torch.cuda.synchronize()
    return a, fn(emu, "cpu")


@pytest.mark.parametrize("R,C,ld", [(7, 64, 64), (300, 768, 768), (33, 96, 128), (256, 3072, 3072)])
def test_split_img_is_the_documented_layout_bit_for_bit(R, C, ld):
    x = rnd(R, ld, seed=1)
    be = ops.HipBackend()
    out = torch.zeros(R, C, device=DEV)
    be.split_img(x.to(DEV), R, C, ld, out, C)
    assert torch.equal(out.cpu().view(torch.int32), img_encode(x[:, :C]).view(torch.int32))
    Rt = (R + 3) // 4 * 4
    xt = rnd(Rt, ld, seed=2)
    ldo = (Rt + 31) // 32 * 32
    outt = torch.zeros(C, ldo, device=DEV)
    be.split_img(xt.to(DEV), Rt, C, ld, outt, ldo, transpose=True)
    ref = torch.zeros(C, ldo)
    ref[:, :Rt] = xt[:, :C].t()
    got = outt.cpu()
    assert torch.equal(img_decode(got)[:, :Rt], img_decode(img_encode(ref))[:, :Rt])


EPIS = [dict(), dict(bias=True), dict(bias=True, act=ops.MM_ACT_GELU_SAVE_GRAD, with_pre=True, keep=True, c_img=True), dict(bias=True, act=ops.MM_ACT_GELU),
        dict(bias=True, aux_mode=ops.MM_AUX_ADD, keep=True), dict(aux_mode=ops.MM_AUX_MUL_LIKE_C, keep=True, c_img=True), dict(act=ops.MM_ACT_RELU, with_pre=True, alpha=0.5)]


@pytest.mark.parametrize("M,N,K", [(600, 768, 768), (257, 160, 64), (1200, 3072, 768), (96, 32, 32), (513, 544, 96), (1, 256, 3072)])
def test_mm_img_nt_epilogues(M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
    bias, aux = rnd(N, seed=3), rnd(M, N, seed=4)
    mask = torch.rand(M, N, generator=torch.Generator().manual_seed(5)) > 0.1
    bits = pack_keep_bits(mask)

    def run(be, dev):
        Ai, Bi = torch.zeros(M, K, device=dev), torch.zeros(N, K, device=dev)
        be.split_img(A.to(dev), M, K, K, Ai, K)
        be.split_img(B.to(dev), N, K, K, Bi, K)
        outs = []
        for e in EPIS:
            kw = {k: v for k, v in e.items() if k in ("act", "aux_mode", "alpha", "c_img")}
            if e.get("bias"):
                kw["bias"] = bias.to(dev)
            if e.get("aux_mode"):
                kw["aux"], kw["ldaux"] = aux.to(dev), N
            if e.get("keep"):
                kw["keep"] = (bits.to(dev), 1.0 / 0.9)
            C = torch.full((M, N), 7.0, device=dev)
            pre = torch.zeros(M, N, device=dev) if e.get("with_pre") else None
            be.mm(MMOp(Ai, K, img=True), MMOp(Bi, K, img=True), C, M, N, K, N, C2=pre, precision=1, **kw)
            outs.append((C, bool(e.get("c_img"))))
            if pre is not None:
                outs.append((pre, False))
        return [o for o, _ in outs] + [torch.tensor([float(f) for _, f in outs])]
    got, ref = both(run)
    flags = ref[-1]
    for g, r, f in zip(got[:-1], ref[:-1], flags):
        if f:
            g, r = img_decode(g), img_decode(r)
        assert rel_l2(g, r) < TOL


def test_mm_img_nt_windows_are_a_strided_convolution():
    """Un-padded windows of an image A (lines_per_seq / seq_stride / ld < K): the conv feature encoder's forward shape."""
    S, Lin, Ci, Co, KH, stride = 3, 45, 64, 96, 3, 2
    Lout = (Lin - KH) // stride + 1
    x, Wg = rnd(S, Lin, Ci, seed=1), rnd(Co, KH * Ci, seed=2, scale=0.1)

    def run(be, dev):
        xi, Wi = torch.zeros(S, Lin, Ci, device=dev), torch.zeros(Co, KH * Ci, device=dev)
        be.split_img(x.to(dev), S * Lin, Ci, Ci, xi, Ci)
        be.split_img(Wg.to(dev), Co, KH * Ci, KH * Ci, Wi, KH * Ci)
        y = torch.zeros(S * Lout, Co, device=dev)
        A = MMOp(xi, stride * Ci, lines_per_seq=Lout, seq_stride=Lin * Ci, seq_elems=Lin * Ci, img=True)
        be.mm(A, MMOp(Wi, KH * Ci, img=True), y, S * Lout, Co, KH * Ci, Co, precision=1)
        return [y]
    got, ref = both(run)
    assert rel_l2(got[0], ref[0]) < TOL
    yd = torch.nn.functional.conv1d(x.double().transpose(1, 2), Wg.double().view(Co, KH, Ci).permute(0, 2, 1), stride=stride).transpose(1, 2)
    assert rel_l2(got[0].view(S, Lout, Co), yd) < 5e-5


@pytest.mark.parametrize("R,M,N,split", [(1000, 768, 768, 0), (96, 64, 32, 1), (4800, 3072, 768, 0), (333, 256, 512, 3), (2000, 2304, 768, 0)])
def test_mm_img_tn_weight_gradient_layout(R, M, N, split):
    """C[m, n] = sum_r A[r, m] B[r, n]: both operands line-contiguous images, split-K slabs folded in fixed order."""
    R4 = (R + 3) // 4 * 4
    A, B = rnd(R4, M, seed=1), rnd(R4, N, seed=2)
    A[R:] = 0
    B[R:] = 0

    def run(be, dev):
        Ai, Bi = torch.zeros(R4, M, device=dev), torch.zeros(R4, N, device=dev)
        be.split_img(A.to(dev), R4, M, M, Ai, M)
        be.split_img(B.to(dev), R4, N, N, Bi, N)
        C = torch.zeros(M, N, device=dev)
        be.mm(MMOp(Ai, M, trans=True, img=True), MMOp(Bi, N, trans=True, img=True), C, M, N, R4, N, split_k=split, precision=1)
        return [C]
    got, ref = both(run)
    assert rel_l2(got[0], ref[0]) < TOL


def test_mm_img_is_bitwise_reproducible_and_rejects_what_it_cannot_do():
    M, N, K = 2048, 768, 768
    A, B = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2).to(DEV)
    be = ops.HipBackend()
    Ai, Bi = torch.zeros_like(A), torch.zeros_like(B)
    be.split_img(A, M, K, K, Ai, K)
    be.split_img(B, N, K, K, Bi, K)
    outs = []
    for _ in range(3):
        C = torch.zeros(M, N, device=DEV)
        be.mm(MMOp(Ai, K, img=True), MMOp(Bi, K, img=True), C, M, N, K, N, precision=1)
        outs.append(C.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    with pytest.raises(a2f_amd.A2FError):
        be.mm(MMOp(Ai, K, img=True), MMOp(B, K), torch.zeros(M, N, device=DEV), M, N, K, N, precision=1)          # mixed formats
    with pytest.raises(a2f_amd.A2FError):
        be.mm(MMOp(Ai, K, img=True), MMOp(Bi, K, img=True), torch.zeros(M, N, device=DEV), M, N, K, N, precision=0)   # bf16x3 only


def test_mm_img_colsum_of_the_stored_values_is_deterministic():
    M, N, K = 1500, 768, 256
    A, B, aux = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(M, N, seed=3)
    mask = torch.rand(M, N, generator=torch.Generator().manual_seed(5)) > 0.1
    bits = pack_keep_bits(mask)

    def run(be, dev):
        Ai, Bi = torch.zeros(M, K, device=dev), torch.zeros(N, K, device=dev)
        be.split_img(A.to(dev), M, K, K, Ai, K)
        be.split_img(B.to(dev), N, K, K, Bi, K)
        outs = []
        for _ in range(2):
            C, cs = torch.zeros(M, N, device=dev), torch.zeros(N, device=dev)
            be.mm(MMOp(Ai, K, img=True), MMOp(Bi, K, img=True), C, M, N, K, N, precision=1, aux=aux.to(dev), aux_mode=ops.MM_AUX_MUL_LIKE_C,
                  keep=(bits.to(dev), 1.25), c_img=True, c_colsum=cs)
            outs += [C, cs]
        return outs
    got, ref = both(run)
    assert torch.equal(got[1], got[3]) and torch.equal(got[0], got[2])
    assert rel_l2(img_decode(got[0]), img_decode(ref[0])) < TOL
    assert rel_l2(got[1], ref[1]) < TOL
    assert rel_l2(got[1], img_decode(got[0]).sum(0)) < 1e-5          # the sums are of the values that were stored


@pytest.mark.parametrize("R,C", [(300, 768), (77, 512), (1030, 1024)])
def test_layernorm_image_outputs(R, C):
    x, g, b, dy = rnd(R, C, seed=1), rnd(C, seed=2), rnd(C, seed=3), rnd(R, C, seed=4)
    mask = torch.rand(R, C, generator=torch.Generator().manual_seed(5)) > 0.1
    bits = pack_keep_bits(mask)

    def run(be, dev):
        y, yi, mu, rs = (torch.zeros(R, C, device=dev), torch.zeros(R, C, device=dev), torch.zeros(R, device=dev), torch.zeros(R, device=dev))
        be.layernorm_fwd_img(x.to(dev), R, C, g.to(dev), b.to(dev), 1e-5, y, yi, mu, rs)
        yi2 = torch.zeros(R, C, device=dev)
        be.layernorm_fwd_img(x.to(dev), R, C, g.to(dev), b.to(dev), 1e-5, None, yi2, mu, rs)
        outs = [y, yi, yi2]
        for keep in (None, (bits.to(dev), 1.0 / 0.9)):
            dx, dxi, dg, db, cs = (torch.zeros(R, C, device=dev), torch.zeros(R, C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                                   torch.zeros(C, device=dev))
            be.layernorm_bwd_img(dy.to(dev), x.to(dev), mu, rs, g.to(dev), R, C, dx, dg, db, dxm_colsum=cs, keep=keep, dxm_img=dxi)
            outs += [dx, dxi, dg, db, cs]
        return outs
    got, ref = both(run)
    assert torch.equal(got[1].view(torch.int32), img_encode(got[0]).view(torch.int32))       # the image IS the split of the fp32 output
    assert torch.equal(got[1], got[2])
    for i, (a, r) in enumerate(zip(got, ref)):
        if i in (1, 2, 4, 9):
            a, r = img_decode(a), img_decode(r)
        assert rel_l2(a, r) < 2e-5, i
    assert torch.equal(got[4].view(torch.int32), img_encode(got[3]).view(torch.int32))       # no mask: image of dx itself
    masked = torch.where(mask, got[8] * (1.0 / 0.9), torch.zeros(()))
    assert rel_l2(img_decode(got[9]), masked) < 1e-5 and rel_l2(got[12], masked.double().sum(0)) < 1e-5


def test_attention_image_outputs_and_image_colsum():
    S, T, H = 2, 150, 12
    ld, ldo = 3 * 64 * H, 64 * H
    qkv, dctx = rnd(S * T, ld, seed=1, scale=0.5), rnd(S * T, ldo, seed=2)
    be = ops.HipBackend()
    q = qkv.to(DEV)
    ctx, ctx_i, lse = torch.zeros(S * T, ldo, device=DEV), torch.zeros(S * T, ldo, device=DEV), torch.zeros(S, H, T, device=DEV)
    be.attn_fwd_img(q, S, T, H, ld, 0.125, ctx, ldo, lse, ctx_i)
    ctx0, lse0 = torch.zeros_like(ctx), torch.zeros_like(lse)
    be.attn_fwd(q, S, T, H, ld, 0.125, ctx0, ldo, lse0)
    # the image holds hi + lo of the same value the fp32 store rounds (the compiler may form the lo residual from the unrounded product: hi + lo is then
    # CLOSER to the exact value than the fp32 word, never further than 2^-16 |x| from it)
    def close(image, ref):
        d = (img_decode(image.cpu()) - ref.cpu().double()).abs()
        return bool((d <= ref.cpu().double().abs() * 2.0 ** -16 + 1e-30).all())
    assert torch.equal(ctx, ctx0) and close(ctx_i, ctx)
    dq_i, dq = torch.zeros(S * T, ld, device=DEV), torch.zeros(S * T, ld, device=DEV)
    be.attn_bwd_img(q, ctx, dctx.to(DEV), lse, S, T, H, ld, ldo, 0.125, dq_i)
    be.attn_bwd(q, ctx, dctx.to(DEV), lse, S, T, H, ld, ldo, 0.125, dq)
    assert close(dq_i, dq)
    cs = torch.zeros(ld, device=DEV)
    be.colsum_img(dq_i, S * T, ld, ld, cs)
    assert rel_l2(cs.cpu(), img_decode(dq_i.cpu()).sum(0)) < 1e-5


@pytest.mark.parametrize("S,L0", [(3, 171), (2, 1000)])
def test_conv_stack_on_image_operands_vs_torch_float64(S, L0, monkeypatch):
    """encoder_ops.ConvStackFn on split-32 images (un-padded forward windows, padded two-row windows + strided image outputs of the gather-form data
    gradient, windowed B of the weight gradient) against torch's conv1d + GELU autograd in float64: output, d/d input, every weight gradient."""
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0)
    C = 64
    strides, ks = (2, 2, 2), (3, 3, 2)
    x = rnd(S, L0, C, seed=1)
    ws = [rnd(C, C, k, seed=10 + i, scale=0.12) for i, k in enumerate(ks)]
    xd = x.double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]
    cur = xd.transpose(1, 2)
    for w, st in zip(wd, strides):
        cur = torch.nn.functional.gelu(torch.nn.functional.conv1d(cur, w, stride=st))
    ref = cur.transpose(1, 2)
    gout = rnd(*ref.shape, seed=5)
    ref.backward(gout.double())
    xg = x.to(DEV).requires_grad_(True)
    wg = [w.to(DEV).requires_grad_(True) for w in ws]
    for x_is_img in (False, True):
        xg.grad = None
        for w in wg:
            w.grad = None
        if x_is_img:
            xi = torch.empty_like(xg)
            ops.backend().split_img(xg.detach(), S * L0, C, C, xi, C)
            y = encoder_ops.ConvStackFn.apply(xi.requires_grad_(True), strides, True, *wg)
        else:
            y = encoder_ops.ConvStackFn.apply(xg, strides, False, *wg)
        assert rel_l2(y, ref) < 5e-5
        y.backward(gout.to(DEV))
        if not x_is_img:
            assert rel_l2(xg.grad, xd.grad) < 5e-5
        for w, r in zip(wg, wd):
            assert rel_l2(w.grad, r.grad) < 5e-5


def test_featenc0_image_output_is_the_split_of_the_fp32_output():
    S, N, C = 2, 3205, 512
    wave, w, g, b = rnd(S, N, seed=1), rnd(C, 10, seed=2, scale=0.3), rnd(C, seed=3), rnd(C, seed=4)
    be = ops.HipBackend()
    L = (N - 10) // 5 + 1
    o32, oi, st = torch.zeros(S, L, C, device=DEV), torch.zeros(S, L, C, device=DEV), torch.zeros(S, 2, C, device=DEV)
    be.featenc0_fwd(wave.to(DEV), S, N, w.to(DEV), g.to(DEV), b.to(DEV), C, 1e-5, o32, st)
    be.featenc0_fwd(wave.to(DEV), S, N, w.to(DEV), g.to(DEV), b.to(DEV), C, 1e-5, oi, st, out_img=True)
    d = (img_decode(oi.cpu()) - o32.cpu().double()).abs()
    assert bool((d <= o32.cpu().double().abs() * 2.0 ** -16 + 1e-30).all())


@pytest.mark.parametrize("T,drop", [(150, False), (150, True), (600, True), (70, True)])
def test_attention_image_inputs_equal_fp32_inputs(T, drop):
    """a2f_attn_fwd_img / a2f_attn_bwd_img with qkv_format / in_format 1 (q|k|v and dctx arrive as the split-32 images the projection GEMMs wrote: tiles and MFMA
    fragments are copied, q is not pre-scaled) against the same kernels fed the fp32 tensors those images stand for: context, log-sum-exp and dqkv agree to the
    rounding of the bf16x3 products (the operands are bit-identical hi / lo pairs; only where the scale 1/8 is applied differs, and it is a power of two)."""
    from emu_backend import img_encode
    S, H = 2, 12
    ld, ldo = 3 * 64 * H, 64 * H
    qkv, dctx = rnd(S * T, ld, seed=11, scale=0.5), rnd(S * T, ldo, seed=12)
    qkv_i, dctx_i = img_encode(qkv), img_encode(dctx)
    qkv_v, dctx_v = img_decode(qkv_i).float(), img_decode(dctx_i).float()          # what the images stand for (differs from qkv below 2^-16 relative)
    be = ops.HipBackend()
    keep_f = keep_b = None
    if drop:
        TW = be.attn_keep_words(T)
        kq = torch.empty(S * H * T, TW, dtype=torch.int32, device=DEV)
        kk = torch.empty(S * H, TW, 32 * TW, dtype=torch.int32, device=DEV)
        be.keep_bits_attn(99, 3, S * H, T, 0.1, kq, kk)
        keep_f, keep_b = (kq, 1.0 / 0.9), (kq, kk, 1.0 / 0.9)
    out = {}
    for tag, q_in, d_in, flag in (("fp32", qkv_v, dctx_v, False), ("img", qkv_i, dctx_i, True)):
        ctx, ctx_i, lse = (torch.zeros(S * T, ldo, device=DEV), torch.zeros(S * T, ldo, device=DEV), torch.zeros(S, H, T, device=DEV))
        be.attn_fwd_img(q_in.to(DEV), S, T, H, ld, 0.125, ctx, ldo, lse, ctx_i, keep=keep_f, qkv_img=flag)
        dq_i = torch.zeros(S * T, ld, device=DEV)
        be.attn_bwd_img(q_in.to(DEV), ctx, d_in.to(DEV), lse, S, T, H, ld, ldo, 0.125, dq_i, keep=keep_b, inputs_img=flag)
        out[tag] = (ctx.cpu(), lse.cpu(), img_decode(dq_i.cpu()))
    for a, b, name in zip(out["img"], out["fp32"], ("ctx", "lse", "dqkv")):
        assert bool(torch.isfinite(a).all()) and rel_l2(a, b) < 2e-5, (name, rel_l2(a, b))
