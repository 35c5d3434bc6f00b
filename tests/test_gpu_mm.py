"""GPU: a2f_mm (bf16-MFMA GEMM family, csrc/gemm_mfma.hip) against the float64 emulator of its contract (include/a2f_hip.h).

precision 1 (bf16x3 split products, the parity setting): <= 2^-16 relative error per product -> 3e-5 relative L2 on the
result; precision 0 (plain bf16 operands, `percision: 16-mixed`): 2^-9 per operand -> 1e-2; precision 2 (bf16x6: three pieces per operand,
the six products down to 2^-16 of the leading one, fp32 accumulate): fp32-GEMM class -> 2e-6.  All calls go through the C ABI.
"""
import pytest
import torch

import a2f_amd
from a2f_amd import ops
from a2f_amd.ops import MMOp
from emu_backend import EmuBackend
from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {1: 3e-5, 0: 1e-2, 2: 2e-6}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def both(fn):
    """fn(be, dev) -> list of output tensors; runs on the HIP backend and on the emulator."""
    hip = ops.HipBackend()
    emu = EmuBackend()
    a = [t.cpu() for t in fn(hip, DEV)]
    torch.cuda.synchronize()
    b = fn(emu, "cpu")
    return a, b


@pytest.mark.parametrize("M,N,K", [(600, 768, 768), (130, 64, 768), (257, 48, 100), (1, 4, 4), (1200, 3072, 768), (96, 132, 36), (333, 200, 4)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("precision", [1, 0, 2])
def test_mm_layouts_epilogues(M, N, K, layout, precision):
    at, bt = layout[0] == "t", layout[1] == "n"
    if at and M % 4:
        M += 4 - M % 4                                     # the contiguous axis must be a multiple of 4 elements
    if bt and N % 4:
        N += 4 - N % 4
    A = rnd(K, M, seed=1) if at else rnd(M, K, seed=1)
    B = rnd(K, N, seed=2) if bt else rnd(N, K, seed=2)
    bias, aux, C0 = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N + 8, seed=5)

    def run(be, dev):
        outs = []
        for kw in (dict(), dict(bias=bias.to(dev)), dict(bias=bias.to(dev), act=ops.MM_ACT_GELU, with_pre=True),
                   dict(act=ops.MM_ACT_RELU, aux=aux.to(dev), aux_mode=ops.MM_AUX_ADD), dict(aux=aux.to(dev), aux_mode=ops.MM_AUX_GELU_BWD),
                   dict(aux=aux.to(dev), aux_mode=ops.MM_AUX_RELU_BWD, accumulate=True, alpha=0.5), dict(split_k=3, bias=bias.to(dev))):
            kw = dict(kw)
            C = C0.clone().to(dev)
            pre = torch.zeros(M, N + 8, device=dev) if kw.pop("with_pre", False) else None
            be.mm(MMOp(A.to(dev), M if at else K, trans=at), MMOp(B.to(dev), N if bt else K, trans=bt), C, M, N, K, N + 8, C2=pre, ldaux=N,
                  precision=precision, **kw)
            outs.append(C)
            if pre is not None:
                outs.append(pre)
        return outs
    got, ref = both(run)
    for g, r in zip(got, ref):
        assert torch.equal(g[:, N:], r[:, N:])                                  # nothing written beyond the N columns
        assert rel_l2(g[:, :N], r[:, :N]) < TOL[precision]


@pytest.mark.parametrize("KH,stride,Lin", [(3, 2, 41), (3, 2, 42), (2, 2, 40), (2, 2, 41), (3, 2, 399)])
@pytest.mark.parametrize("S", [1, 3])
def test_mm_conv1d_composers_vs_torch(KH, stride, Lin, S):
    """Conv1d forward / data gradient / weight gradient composed from a2f_mm windows against torch's conv1d in float64."""
    Ci, Co = 64, 96
    Lout = (Lin - KH) // stride + 1
    x, W, dy = rnd(S, Lin, Ci, seed=1), rnd(Co, Ci, KH, seed=2, scale=0.1), rnd(S, Lout, Co, seed=3)
    Wg = W.permute(0, 2, 1).reshape(Co, KH * Ci).contiguous()
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    ref = torch.nn.functional.conv1d(xd.transpose(1, 2), Wd, stride=stride).transpose(1, 2)
    ref.backward(dy.double())
    be = ops.HipBackend()
    y = torch.empty(S, Lout, Co, device=DEV)
    ops.mm_conv_fwd(be, x.to(DEV), Wg.to(DEV), y, S, Lin, Lout, Ci, Co, KH, stride)
    assert rel_l2(y, ref) < 3e-5
    dx = torch.zeros(S, Lin, Ci, device=DEV)
    ops.mm_conv_dgrad(be, dy.to(DEV), Wg.to(DEV), dx, S, Lin, Lout, Ci, Co, KH, stride)
    assert rel_l2(dx, xd.grad) < 3e-5
    dWg = torch.empty(Co, KH * Ci, device=DEV)
    ops.mm_conv_wgrad(be, x.to(DEV), dy.to(DEV), dWg, S, Lin, Lout, Ci, Co, KH, stride)
    assert rel_l2(dWg, Wd.grad.permute(0, 2, 1).reshape(Co, KH * Ci)) < 3e-5


def test_mm_batched_groups_and_split_k():
    """The grouped positional convolution's shape: 16 independent problems (batch strides), N = 48, K = 6144, windows with
    left/right zero padding; and a weight-gradient-shaped contraction over 9600 rows with automatic split-K."""
    G, T, Cg, KH = 16, 100, 48, 128
    pad = KH // 2
    xg = rnd(G, T, Cg, seed=1)                       # group-major activations (G, T, Cg)
    wg = rnd(G, Cg, KH * Cg, seed=2, scale=0.02)     # (G, Co_g, KH*Cg)
    bias = rnd(G * Cg, seed=3)

    def run(be, dev):
        y = torch.zeros(T, G * Cg, device=dev)
        A = MMOp(xg.to(dev), Cg, lines_per_seq=T, seq_stride=T * Cg, off=-pad * Cg, seq_elems=T * Cg)
        be.mm(A, MMOp(wg.to(dev), KH * Cg), y, T, Cg, KH * Cg, G * Cg, bias=bias.to(dev), act=ops.MM_ACT_GELU, batch=G, a_bstride=T * Cg,
              b_bstride=Cg * KH * Cg, c_bstride=Cg, bias_bstride=Cg)
        dy, x = rnd(9600, 256, seed=4).to(dev), rnd(9600, 768, seed=5).to(dev)
        dW = torch.empty(256, 768, device=dev)
        ops.mm_linear_wgrad(be, dy, x, dW, 9600, 256, 768)
        return [y, dW]
    got, ref = both(run)
    for g, r in zip(got, ref):
        assert rel_l2(g, r) < 3e-5


def test_mm_rejects_bad_arguments():
    be = ops.HipBackend()
    x = torch.zeros(8, 6, device=DEV)
    with pytest.raises(a2f_amd.A2FError):
        be.mm(MMOp(x, 6), MMOp(x, 6), torch.zeros(8, 8, device=DEV), 8, 8, 6, 8)       # K = 6 is not a multiple of 4


def test_mm_saved_gelu_derivative_epilogues():
    """act 4 (GELU with C2 = gelu'(v)) and aux_mode 4 (multiply by aux at C's own index, row map and c_off included): a forward
    Linear + GELU that saves its derivative, and a strided-convolution data gradient (two launches writing disjoint positions)
    multiplied by the derivative saved by the layer below, against the float64 emulator."""
    from a2f_amd.ops import MM_ACT_GELU_SAVE_GRAD, mm_conv_dgrad
    M, N, K = 300, 260, 96
    x, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)

    def fwd(be, dev):
        y, gp = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
        be.mm(MMOp(x.to(dev), K), MMOp(W.to(dev), K), y, M, N, K, N, bias=b.to(dev), act=MM_ACT_GELU_SAVE_GRAD, C2=gp)
        return [y, gp]
    a, e = both(fwd)
    assert rel_l2(a[0], e[0]) < 3e-5 and rel_l2(a[1], e[1]) < 3e-5
    z = (x.double() @ W.double().t() + b.double())
    zt = z.clone().requires_grad_(True)
    torch.nn.functional.gelu(zt).sum().backward()
    assert rel_l2(a[1], zt.grad) < 3e-5                        # C2 really is d gelu / d pre-activation
    S, Lin, Ci, Co, KH, stride = 2, 37, 8, 12, 3, 2
    Lout = (Lin - KH) // stride + 1
    dy, wg, mul = rnd(S, Lout, Co, seed=4), rnd(Co, KH * Ci, seed=5, scale=0.3), rnd(S, Lin, Ci, seed=6)

    def dgrad(be, dev):
        dx = torch.zeros(S, Lin, Ci, device=dev)
        mm_conv_dgrad(be, dy.to(dev), wg.to(dev), dx, S, Lin, Lout, Ci, Co, KH, stride, mul=mul.to(dev))
        plain = torch.zeros(S, Lin, Ci, device=dev)
        mm_conv_dgrad(be, dy.to(dev), wg.to(dev), plain, S, Lin, Lout, Ci, Co, KH, stride)
        return [dx, plain]
    a, e = both(dgrad)
    assert rel_l2(a[0], e[0]) < 3e-5
    assert rel_l2(a[0], a[1] * mul) < 1e-6


@pytest.mark.parametrize("M,N,K,split", [(768, 768, 19200, 0), (3072, 768, 9600, 0), (132, 260, 777, 1), (768, 200, 4100, 5), (64, 48, 96, 1)])
@pytest.mark.parametrize("precision", [1, 0, 2])
def test_mm_weight_gradient_with_fused_bias_gradient(M, N, K, split, precision):
    """a_rowsum: dW = dY^T X and db = column sums of dY out of ONE kernel (plain path: K a multiple of 32; masked path otherwise;
    with and without split-K slabs), against the float64 emulator; the sums are fp32 additions of the fp32 operand, so db holds the
    tighter bound of a plain reduction in both precisions."""
    dy, x = rnd(K, M, seed=1), rnd(K, N, seed=2)

    def run(be, dev):
        dW, db = torch.zeros(M, N, device=dev), torch.full((M,), 7.0, device=dev)
        be.mm(MMOp(dy.to(dev), M, trans=True), MMOp(x.to(dev), N, trans=True), dW, M, N, K, N, split_k=split, precision=precision, a_rowsum=db)
        return [dW, db]
    a, e = both(run)
    assert rel_l2(a[0], e[0]) < TOL[precision]
    assert rel_l2(a[1], e[1]) < 1e-5


@pytest.mark.parametrize("M,N,K", [(1200, 3072, 768), (1100, 2304, 768), (5000, 3072, 768), (700, 768, 3072)])
def test_mm_panel_major_tile_order_is_bitwise_the_row_major_result(M, N, K):
    """The tile order (row-major, or panels of n-tiles when the B operand outgrows an XCD's L2: gemm_mfma.h tile_origin) only moves tiles
    between workgroups: the Linear forward with the layers' epilogues gives bit-identical results under both (A2F_MM_GROUP_N=0 forces
    row-major; read once per process, hence the child process)."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import a2f_amd; from a2f_amd import ops; from a2f_amd.ops import MMOp, MM_ACT_GELU_SAVE_GRAD\n"
        "g = torch.Generator().manual_seed(5); M, N, K = %d, %d, %d\n"
        "x, W, b = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * 0.3).cuda(), torch.randn(N, generator=g).cuda()\n"
        "y, z = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')\n"
        "ops.backend().mm(MMOp(x, K), MMOp(W, K), y, M, N, K, N, bias=b, act=MM_ACT_GELU_SAVE_GRAD, C2=z, precision=1)\n"
        "torch.save((y.cpu(), z.cpu()), sys.argv[1])\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), M, N, K)
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for i, env_extra in enumerate(({}, {"A2F_MM_GROUP_N": "0"})):
            path = os.path.join(d, f"o{i}.pt")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-1500:]
            outs.append(torch.load(path))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.functional.gelu(torch.nn.functional.linear(*[t for t in _panel_inputs(M, N, K)]))
    assert rel_l2(outs[0][0], ref) < 1e-4


def test_mm_pointer_epilogue_is_bitwise_the_buffer_epilogue():
    """The epilogue addresses C / C2 / aux / keep words through buffer resources with 32-bit offsets when every span is below 2 GiB and
    through 64-bit pointers otherwise (gemm_mfma.h tile_epilogue_band, Args.epi32).  A2F_MM_EPI32=0 forces the pointer form (read once
    per process): every epilogue variant, ragged tile edges, a padded ldc, row-sequence output and split-K give the same bits."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import a2f_amd; from a2f_amd import ops; from a2f_amd.ops import MMOp\n"
        "g = torch.Generator().manual_seed(9); M, N, K = 333, 200, 96\n"
        "r = lambda *s: torch.randn(*s, generator=g).cuda()\n"
        "A, B, bias, aux, like = r(M, K), r(N, K), r(N), r(M, N), r(M, N + 8)\n"
        "keep = torch.randint(0, 2**31 - 1, (M, (N + 31) // 32), generator=g, dtype=torch.int32).cuda()\n"
        "outs = []\n"
        "for kw in (dict(), dict(bias=bias, act=ops.MM_ACT_GELU_SAVE_GRAD, C2=True), dict(bias=bias, act=ops.MM_ACT_GELU, C2=True),\n"
        "           dict(act=ops.MM_ACT_RELU, aux=aux, aux_mode=ops.MM_AUX_ADD), dict(aux=aux, aux_mode=ops.MM_AUX_GELU_BWD, accumulate=True, alpha=0.5),\n"
        "           dict(aux=aux, aux_mode=ops.MM_AUX_RELU_BWD), dict(split_k=3, bias=bias), dict(bias=bias, keep=(keep, 1.25)),\n"
        "           dict(aux=like, aux_mode=ops.MM_AUX_MUL_LIKE_C), dict(c_rows_per_seq=37, c_seq_stride=37 * (N + 8) + 16, big=True)):\n"
        "    kw = dict(kw); C = torch.zeros(9 * (37 * (N + 8) + 16) + 8, device='cuda') if kw.pop('big', False) else like.clone(); C2 = torch.zeros(M, N + 8, device='cuda') if kw.pop('C2', False) else None\n"
        "    ops.backend().mm(MMOp(A, K), MMOp(B, K), C, M, N, K, N + 8, C2=C2, ldaux=N, precision=1, **kw)\n"
        "    outs += [C.cpu()] + ([C2.cpu()] if C2 is not None else [])\n"
        "torch.save(outs, sys.argv[1])\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for i, env_extra in enumerate(({}, {"A2F_MM_EPI32": "0"})):
            path = os.path.join(d, f"o{i}.pt")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-1500:]
            outs.append(torch.load(path))
    assert len(outs[0]) == len(outs[1]) == 12
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _panel_inputs(M, N, K):
    g = torch.Generator().manual_seed(5)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.3, torch.randn(N, generator=g)
    return x.double(), W.double(), b.double()
