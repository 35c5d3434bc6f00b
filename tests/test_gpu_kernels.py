"""GPU: every HIP entry point of liba2f_hip.so against the CPU emulator of its contract (include/a2f_hip.h).

All calls go through the C ABI (ctypes) exactly as the product does.  fp32 tolerance: the f32 MFMA is an exact fp32
fma chain, so only summation order differs from the float64 emulator -> 2e-5 relative L2 is ample; the head kernel's
default configuration evaluates its products as bf16x3 split sums (hi*hi + hi*lo + lo*hi, fp32 accumulate: <= 2^-16 per
product) and stays inside the same 2e-5.  Parity against the reference goldens is in test_gpu_models.py at the 1e-4 bar
of BASELINE.json.
"""
import math
import os
import sys

import pytest
import torch

import a2f_amd
from a2f_amd import ops
from emu_backend import EmuBackend
from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def run_gemm_both(A, B, Cinit, M, N, K, a, b, ldc, precision=0, **kw):
    hip, emu = ops.HipBackend(), EmuBackend()
    hip.gemm_precision = precision
    outs = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        kk = dict(kw)
        for name in ("bias", "aux", "ones_out"):
            if kk.get(name) is not None:
                kk[name] = kk[name].clone().to(dev)
        Cc = Cinit.clone().to(dev)
        be.gemm(A.to(dev), B.to(dev), Cc, M, N, K, a, b, ldc, **kk)
        outs.append((Cc.cpu(), kk["ones_out"].cpu() if kk.get("ones_out") is not None else None))
    return outs


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 50, 50), (300, 72, 111), (513, 129, 67), (200, 96, 192), (1000, 243, 486), (64, 15, 3)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("precision,tol", [(1, 3e-5), (2, 2e-6)])
def test_gemm_split_precision_layouts_and_epilogues(M, N, K, layout, precision, tol):
    """a2f_gemm_desc.precision = 1 (bf16x3: three bf16 MFMAs per product, 2^-16) and 2 (bf16x6: six MFMAs, fp32-class)
    against the float64 contract: same operand layouts, masks, epilogues and split-K slabs as the exact kernel."""
    if layout == "nt":
        A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
        a, b = (0, K, 1), (0, 1, K)
    elif layout == "nn":
        A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
        a, b = (0, K, 1), (0, N, 1)
    else:
        A, B = rnd(K, M, seed=1), rnd(K, N, seed=2)
        a, b = (0, 1, M), (0, N, 1)
    bias, aux, C0 = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N + 3, seed=5)
    for kw in (dict(), dict(bias=bias, act=1), dict(aux=aux, ldaux=N, aux_mode=2, accumulate=True), dict(bias=bias, aux=aux, ldaux=N, aux_mode=3),
               dict(mask=(1, 4, 1, 0, min(3, K))), dict(mask=(2, 5, 0, 0, min(7, N)))):
        (c_hip, _), (c_emu, _) = run_gemm_both(A, B, C0, M, N, K, a, b, N + 3, precision=precision, **kw)
        assert rel_l2(c_hip[:, :N], c_emu[:, :N]) < tol, kw.keys()
        assert torch.equal(c_hip[:, N:], C0[:, N:])
    if K >= 50:                                                  # deterministic split-K slabs + ones column
        ws = torch.empty(3 * M * (N + 1), device=DEV)
        hip = ops.HipBackend(); hip.gemm_precision = precision
        Cc, ones = torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV)
        hip.gemm(A.to(DEV), B.to(DEV), Cc, M, N, K, a, b, N, split_k=3, ones_out=ones, splitk_ws=ws)
        ref = (A.double() if layout != "tn" else A.double().t()) @ (B.double().t() if layout == "nt" else B.double())
        assert rel_l2(Cc.cpu(), ref) < tol
        asum = (A.double() if layout != "tn" else A.double().t()).sum(1)
        assert rel_l2(ones.cpu(), asum) < tol


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 50, 50), (128, 32, 16), (300, 72, 111), (513, 129, 67), (200, 96, 192), (64, 15, 3)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_plain_layouts(M, N, K, layout):
    if layout == "nt":      # A (M,K) row-major, B = W (N,K) row-major (torch Linear)
        A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
        a, b = (0, K, 1), (0, 1, K)
    elif layout == "nn":    # B (K,N) row-major
        A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
        a, b = (0, K, 1), (0, N, 1)
    else:                   # A stored (K,M) row-major
        A, B = rnd(K, M, seed=1), rnd(K, N, seed=2)
        a, b = (0, 1, M), (0, N, 1)
    (c_hip, _), (c_emu, _) = run_gemm_both(A, B, torch.zeros(M, N), M, N, K, a, b, N)
    assert rel_l2(c_hip, c_emu) < 2e-5


@pytest.mark.parametrize("act,aux_mode", [(0, 0), (1, 0), (2, 0), (0, 1), (0, 2), (0, 3), (1, 3)])
def test_gemm_epilogues(act, aux_mode):
    M, N, K = 211, 77, 45
    A, B = rnd(M, K, seed=3), rnd(N, K, seed=4)
    bias = rnd(N, seed=5)
    aux = rnd(M, N + 3, seed=6) if aux_mode else None
    C0 = rnd(M, N + 5, seed=7)
    for accumulate in (False, True):
        (c_hip, _), (c_emu, _) = run_gemm_both(A, B, C0, M, N, K, (0, K, 1), (0, 1, K), N + 5, bias=bias, act=act, aux=aux,
                                               ldaux=N + 3, aux_mode=aux_mode, accumulate=accumulate)
        assert rel_l2(c_hip, c_emu) < 2e-5
        assert torch.equal(c_hip[:, N:], C0[:, N:])          # columns beyond N untouched (ldc > N)


def test_gemm_broadcast_aux_row_and_c_offset():
    M, N, K = 70, 40, 20
    A, B = rnd(M, K, seed=8), rnd(N, K, seed=9)
    aux = rnd(1, N, seed=10)
    C0 = torch.zeros(M + 1, N)
    (c_hip, _), (c_emu, _) = run_gemm_both(A, B, C0, M, N, K, (0, K, 1), (0, 1, K), N, c_off=N, aux=aux, ldaux=0, aux_mode=3)
    assert rel_l2(c_hip, c_emu) < 2e-5 and float(c_hip[0].abs().sum()) == 0.0


@pytest.mark.parametrize("B_,P_out,Cin,Co", [(5, 8, 37, 32), (3, 4, 32, 32), (7, 2, 32, 64), (6, 1, 64, 64), (2, 16, 1, 72), (3, 2, 243, 256)])
def test_conv_implicit_gemm_fwd_wgrad_dgrad(B_, P_out, Cin, Co):
    """k=3/s=2/p=1 convolution over channels-last rows == torch conv1d, forward, weight grad and data grad."""
    P_in = 2 * P_out
    act = rnd(B_, P_in, Cin, seed=11).abs() * torch.sign(rnd(B_, P_in, Cin, seed=12))      # mixed signs for the ReLU mask
    w = rnd(Co, Cin, 3, seed=13, scale=0.2)
    bias = rnd(Co, seed=14)
    x = act.permute(0, 2, 1).clone().requires_grad_(True)                      # (B, Cin, P_in)
    wt = w.clone().requires_grad_(True)
    bt = bias.clone().requires_grad_(True)
    y = torch.relu(torch.nn.functional.conv1d(x, wt, bt, stride=2, padding=1))   # (B, Co, P_out)
    dy = rnd(B_, P_out, Co, seed=15)
    y.backward(dy.permute(0, 2, 1))
    be = ops.HipBackend()
    wg = torch.empty(Co, 3 * Cin, device=DEV)
    be.conv_weight_pack(w.to(DEV), wg, Co, Cin, 3, 0)
    assert torch.equal(wg.cpu().view(Co, 3, Cin), w.permute(0, 2, 1))
    out = torch.empty(B_, P_out, Co, device=DEV)
    R = B_ * P_out
    ops.conv_fwd(be, act.to(DEV), wg, bias.to(DEV), out, R, P_out, Cin, Co)
    assert rel_l2(out.cpu(), y.detach().permute(0, 2, 1)) < 2e-5
    d_pre = (dy * (out.cpu() > 0)).to(DEV).contiguous()                       # gradient w.r.t. the pre-ReLU output
    dwg = torch.zeros(Co, 3 * Cin, device=DEV)
    dbias = torch.zeros(Co, device=DEV)
    ops.conv_bwd_weight(be, act.to(DEV), d_pre, dwg, dbias, R, P_out, Cin, Co)
    dw = torch.empty(Co, Cin, 3, device=DEV)
    be.conv_weight_pack(dwg, dw, Co, Cin, 3, 1)
    assert rel_l2(dw.cpu(), wt.grad) < 5e-5
    assert rel_l2(dbias.cpu(), bt.grad) < 5e-5
    d_in = torch.full((B_, P_in, Cin), float("nan"), device=DEV)
    ops.conv_k3s2_bwd_input(be, d_pre, wg, d_in, R, P_out, Cin, Co)
    assert rel_l2(d_in.cpu(), x.grad.permute(0, 2, 1)) < 5e-5
    # with the ReLU mask of the layer below fused in
    d_in2 = torch.full((B_, P_in, Cin), float("nan"), device=DEV)
    ops.conv_k3s2_bwd_input(be, d_pre, wg, d_in2, R, P_out, Cin, Co, relu_of=act.to(DEV))
    assert rel_l2(d_in2.cpu(), x.grad.permute(0, 2, 1) * (act > 0)) < 5e-5


@pytest.mark.parametrize("M,N,K", [(128, 256, 768), (500, 243, 486), (37, 50, 300)])
def test_gemm_slab_splitk_with_every_epilogue(M, N, K):
    """Deterministic slab split-K is valid with the full epilogue (bias, activation, aux modes, accumulate): the fold kernel
    applies it.  Explicit split_k and the host layer's automatic choice for under-filled grids."""
    A, B = rnd(M, K, seed=70), rnd(N, K, seed=71)
    bias, aux, C0 = rnd(N, seed=72), rnd(M, N, seed=73), rnd(M, N + 3, seed=74)
    for kw in (dict(bias=bias, act=1), dict(bias=bias, act=2), dict(aux=aux, ldaux=N, aux_mode=1), dict(aux=aux, ldaux=N, aux_mode=2, accumulate=True),
               dict(bias=bias, aux=aux, ldaux=N, aux_mode=3)):
        for split in (1, 5):                                     # 1: automatic (K >= 256, few tiles), 5: explicit
            kk = dict(kw)
            if split > 1:
                kk.update(split_k=split, splitk_ws=torch.empty(split * M * (N + 1), device=DEV))
            hip, emu = ops.HipBackend(), EmuBackend()
            outs = []
            for be, dev in ((hip, DEV), (emu, "cpu")):
                k2 = {k: (v.clone().to(dev) if torch.is_tensor(v) and k != "splitk_ws" else v) for k, v in kk.items()}
                if dev == "cpu":
                    k2.pop("splitk_ws", None); k2.pop("split_k", None)
                Cc = C0.clone().to(dev)
                be.gemm(A.to(dev), B.to(dev), Cc, M, N, K, (0, K, 1), (0, 1, K), N + 3, **k2)
                outs.append(Cc.cpu())
            assert rel_l2(outs[0][:, :N], outs[1][:, :N]) < 2e-5, (kw.keys(), split)
            assert torch.equal(outs[0][:, N:], C0[:, N:])


def test_gemm_splitk_ones_column():
    M, N, K = 50, 128, 5000
    dy, x = rnd(K, M, seed=16), rnd(K, N, seed=17)
    dW = torch.zeros(M, N)
    db = torch.zeros(M)
    for split in (1, 7, 64):
        (c_hip, o_hip), (c_emu, o_emu) = run_gemm_both(dy, x, dW, M, N, K, (0, 1, M), (0, N, 1), N, ones_out=db, split_k=split)
        assert rel_l2(c_hip, dy.double().t() @ x.double()) < 2e-5
        assert rel_l2(o_hip, dy.double().sum(0)) < 2e-5
        assert rel_l2(c_hip, c_emu) < 2e-5


def test_gemm_rejects_bad_arguments():
    be = ops.HipBackend()
    A = torch.zeros(4, 4, device=DEV)
    with pytest.raises(a2f_amd.A2FError):
        be.gemm(A, A, A, 0, 4, 4, (0, 4, 1), (0, 1, 4), 4)
    with pytest.raises(a2f_amd.A2FError):
        be.gemm(A, A, A, 4, 4, 4, (0, 4, 1), (0, 1, 4), 2)            # ldc < N
    with pytest.raises(a2f_amd.A2FError):
        be.gemm(A, A, A, 4, 4, 4, (0, 4, 1), (0, 1, 4), 4, split_k=2, act=1)
    with pytest.raises(a2f_amd.A2FError):
        be.gemm(A.cpu(), A, A, 4, 4, 4, (0, 4, 1), (0, 1, 4), 4)      # CPU tensor: no CPU path


@pytest.mark.parametrize("B_,n", [(2, 33), (6, 15069), (64, 15069), (10, 3)])
def test_vocaloss_fwd_bwd(B_, n):
    pred, gt = rnd(B_, n, seed=20, scale=3.0), rnd(B_, n, seed=21, scale=3.0)
    g3 = torch.tensor([1.0, 0.25, -0.5])
    hip, emu = ops.HipBackend(), EmuBackend()
    o_h, o_e = torch.empty(3, device=DEV), torch.empty(3)
    hip.vocaloss_fwd(pred.to(DEV), gt.to(DEV), B_, n, 1.0, 10.0, o_h)
    emu.vocaloss_fwd(pred, gt, B_, n, 1.0, 10.0, o_e)
    assert torch.allclose(o_h.cpu(), o_e, rtol=2e-6, atol=0)
    g_h, g_e = torch.empty(B_, n, device=DEV), torch.empty(B_, n)
    hip.vocaloss_bwd(pred.to(DEV), gt.to(DEV), g3.to(DEV), B_, n, 1.0, 10.0, g_h)
    emu.vocaloss_bwd(pred, gt, g3, B_, n, 1.0, 10.0, g_e)
    assert rel_l2(g_h.cpu(), g_e) < 1e-6
    # determinism of the forward reduction
    o2 = torch.empty(3, device=DEV)
    hip.vocaloss_fwd(pred.to(DEV), gt.to(DEV), B_, n, 1.0, 10.0, o2)
    assert torch.equal(o2, o_h)


def test_vocaloss_rejects_odd_batch():
    be = ops.HipBackend()
    p = torch.zeros(3, 6, device=DEV)
    with pytest.raises(a2f_amd.A2FError):
        be.vocaloss_fwd(p, p, 3, 6, 1.0, 10.0, torch.empty(3, device=DEV))


def test_pack_kernels():
    hip, emu = ops.HipBackend(), EmuBackend()
    B_ = 9
    x, oh = rnd(B_, 29, 16, seed=30), rnd(B_, 12, seed=31)
    a_h, c_h = torch.zeros(B_, 16, 37, device=DEV), torch.zeros(B_, 72, device=DEV)
    a_e, c_e = torch.zeros(B_, 16, 37), torch.zeros(B_, 72)
    hip.voca_pack(x.to(DEV), oh.to(DEV), B_, 12, a_h, c_h)
    emu.voca_pack(x, oh, B_, 12, a_e, c_e)
    assert torch.equal(a_h.cpu(), a_e) and torch.equal(c_h.cpu(), c_e)
    x2 = rnd(B_, 52, 32, seed=32)
    a_h, c_h = torch.zeros(B_, 64, 32, device=DEV), torch.zeros(B_, 268, device=DEV)
    a_e, c_e = torch.zeros(B_, 64, 32), torch.zeros(B_, 268)
    hip.a2m_pack(x2.to(DEV), oh.to(DEV), B_, 12, a_h, c_h, 268, 256)
    emu.a2m_pack(x2, oh, B_, 12, a_e, c_e, 268, 256)
    assert torch.equal(a_h.cpu(), a_e) and torch.equal(c_h.cpu(), c_e)


def test_adam_matches_torch_optim():
    n = 100003
    p0, g = rnd(n, seed=40), rnd(n, seed=41)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-4, weight_decay=1e-5)
    be = ops.HipBackend()
    npad = (n + 3) // 4 * 4
    p = torch.zeros(npad, device=DEV); p[:n] = p0.to(DEV)
    gg = torch.zeros(npad, device=DEV)
    m, v = torch.zeros(npad, device=DEV), torch.zeros(npad, device=DEV)
    for step in range(1, 4):
        gs = g * step
        ref.grad = gs.clone()
        opt.step()
        gg[:n] = gs.to(DEV)
        be.adam_step(p, gg, m, v, n, 1e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0, step)
        assert rel_l2(p[:n].cpu() - p0, ref.detach() - p0) < 1e-5


@pytest.mark.parametrize("B_,n,K,trows,Bloss", [(4, 15069, 50, 4, 4), (70, 1536, 50, 70, 70), (64, 999, 64, 1, 64), (33, 600, 64, 1, 32),
                                                 (2, 3, 50, 2, 2), (130, 15069, 64, 1, 130), (256, 15069, 50, 256, 256),
                                                 (601, 15069, 64, 1, 600), (1031, 4002, 50, 1031, 1030),
                                                 # one template row per sequence (FaceFormer: S sequences of T frames in one launch)
                                                 (120, 15069, 64, 3, 120), (1200, 4002, 64, 2, 1200), (96, 15069, 50, 6, 96), (38, 999, 64, 19, 38),
                                                 (130, 1536, 64, 5, 104),
                                                 # ... several row blocks per workgroup AND template rows that change inside a block, full width
                                                 (216, 15069, 64, 9, 216), (240, 15069, 64, 8, 240), (1200, 15069, 64, 2, 1200)])
def test_head_fused_vs_contract(B_, n, K, trows, Bloss):
    hip, emu = ops.HipBackend(), EmuBackend()
    H, Wt, bias = rnd(B_, K, seed=50), rnd(n, K, seed=51, scale=0.3), rnd(n, seed=52)
    templ = rnd(trows, n, seed=53, scale=5.0)
    gt = templ.repeat_interleave(B_ // trows, dim=0) + rnd(B_, n, seed=54, scale=0.5)
    outs = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        pred = torch.full((B_, n), float("nan"), device=dev)
        out3 = torch.empty(3, device=dev)
        dH, dW, db = torch.full((B_, K), float("nan"), device=dev), torch.full((n, K), float("nan"), device=dev), torch.full((n,), float("nan"), device=dev)
        be.head_fused(H.to(dev), Wt.to(dev), bias.to(dev), templ.to(dev), trows, gt.contiguous().to(dev), B_, Bloss, n, K, 1.0, 10.0,
                      pred, out3, dH, dW, db)
        outs.append([t.cpu() for t in (pred, out3, dH, dW, db)])
    (p_h, o_h, dH_h, dW_h, db_h), (p_e, o_e, dH_e, dW_e, db_e) = outs
    assert rel_l2(p_h, p_e) < 2e-6
    # the tile rows are fetched as 16-byte pieces that may straddle the row end: the very last elements must still be exact-class
    assert torch.allclose(p_h[-1, -8:], p_e[-1, -8:], rtol=1e-4, atol=1e-4) and torch.allclose(p_h[:, -1], p_e[:, -1], rtol=1e-4, atol=1e-4)
    assert torch.allclose(o_h, o_e, rtol=2e-5, atol=0)
    assert rel_l2(dH_h, dH_e) < 2e-5
    assert rel_l2(dW_h, dW_e) < 2e-5
    assert rel_l2(db_h, db_e) < 2e-5


def test_head_fused_rejects_bad_arguments():
    be = ops.HipBackend()
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(a2f_amd.A2FError):       # unsupported K
        be.head_fused(z(4, 32), z(30, 32), z(30), z(4, 30), 4, z(4, 30), 4, 4, 30, 32, 1.0, 10.0, z(4, 30), z(3), z(4, 32), z(30, 32), z(30))
    with pytest.raises(a2f_amd.A2FError):       # template rows must divide B
        be.head_fused(z(4, 50), z(30, 50), z(30), z(3, 30), 3, z(4, 30), 4, 4, 30, 50, 1.0, 10.0, z(4, 30), z(3), z(4, 50), z(30, 50), z(30))
    with pytest.raises(a2f_amd.A2FError):       # odd Bloss
        be.head_fused(z(4, 50), z(30, 50), z(30), z(4, 30), 4, z(4, 30), 4, 3, 30, 50, 1.0, 10.0, z(4, 30), z(3), z(4, 50), z(30, 50), z(30))


@pytest.mark.parametrize("R,C", [(7, 3), (1000, 72), (4096, 108), (333, 162), (2049, 243), (512, 256), (5, 1000), (333, 164), (2049, 244), (70001, 72), (3, 8)])
@pytest.mark.parametrize("training,relu", [(True, True), (True, False), (False, True)])
def test_batchnorm_fwd_bwd(R, C, training, relu):
    hip, emu = ops.HipBackend(), EmuBackend()
    x = rnd(R, C, seed=60, scale=2.0) + 0.5
    gamma, beta = rnd(C, seed=61) * 0.1 + 1.0, rnd(C, seed=62) * 0.1
    rm0, rv0 = rnd(C, seed=63) * 0.1, rnd(C, seed=64).abs() + 0.5
    dy = rnd(R, C, seed=65)
    res = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        rm, rv = rm0.clone().to(dev), rv0.clone().to(dev)
        y, mu, istd = torch.empty(R, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
        be.bn_fwd(x.to(dev), R, C, gamma.to(dev), beta.to(dev), rm, rv, 0.1, 1e-5, training, relu, y, mu, istd)
        dx, dg, dbt = torch.empty(R, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
        be.bn_bwd(x.to(dev), beta.to(dev), dy.to(dev), R, C, gamma.to(dev), mu, istd, training, relu, dx, dg, dbt)
        res.append([t.cpu() for t in (y, rm, rv, dx, dg, dbt)])
    for name, a, b in zip(("y", "running_mean", "running_var", "dx", "dgamma", "dbeta"), res[0], res[1]):
        assert rel_l2(a, b) < 2e-5, name


_FF_SPEC = {}


@pytest.mark.parametrize("S,T,period", [(1, 5, 60), (3, 70, 60), (2, 33, 7), (2, 1, 7), (2, 300, 60), (1, 257, 7), (9, 24, 60), (8, 30, 60), (16, 40, 60)])
def test_ff_decoder_kernels_vs_spec(S, T, period):
    """Recurrence forward + reverse-time backward against autograd on the float64 statement of the same recurrence.  The backward is the record
    form (recurrence kernel + weight-gradient products over all frames; from 129 frames on in blocks of 128 steps with the cross-block attention terms batched over
    the chip: T = 300 / 257 run three blocks)."""
    from a2f_amd._lib import load as _load
    nP = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64 + 6 * 64 + 64 * 64 + 64
    P = rnd(nP, seed=70, scale=0.12)
    o_ln = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64
    P[o_ln:o_ln + 64] += 1.0; P[o_ln + 128:o_ln + 192] += 1.0; P[o_ln + 256:o_ln + 320] += 1.0       # LayerNorm gains near 1
    P[o_ln + 384:o_ln + 384 + 64 * 64] *= 0.3                                                           # contractive feedback
    pe = rnd(period, 64, seed=71)
    cross, style, dy = rnd(S, T, 64, seed=72), rnd(S, 64, seed=73), rnd(S, T, 64, seed=74)
    hip, emu = ops.HipBackend(), EmuBackend()
    y_h = torch.empty(S, T, 64, device=DEV)
    save = torch.empty(hip.ff_save_floats(S, T), device=DEV)
    hip.ff_decoder_fwd(P.to(DEV), pe.to(DEV), cross.to(DEV), style.to(DEV), S, T, period, y_h, save)
    key = ("plain", S, T, period)
    if key not in _FF_SPEC:                                      # the float64 autograd statement is the slow part: once per shape, shared by the forms
        y_e = torch.empty(S, T, 64)
        emu.ff_decoder_fwd(P, pe, cross, style, S, T, period, y_e, None)
        dP_e, dc_e, ds_e = torch.zeros(nP), torch.empty(S, T, 64), torch.empty(S, 64)
        emu.ff_decoder_bwd(P, dy, S, T, period, None, dP_e, dc_e, ds_e)
        _FF_SPEC[key] = (y_e, dP_e, dc_e, ds_e)
    y_e, dP_e, dc_e, ds_e = _FF_SPEC[key]
    assert rel_l2(y_h.cpu(), y_e) < 2e-5
    dP_h, dc_h, ds_h = torch.full((nP,), 7.0, device=DEV), torch.empty(S, T, 64, device=DEV), torch.empty(S, 64, device=DEV)   # dparams is OVERWRITTEN
    hip.ff_decoder_bwd(P.to(DEV), dy.to(DEV), S, T, period, save, dP_h, dc_h, ds_h)
    # deterministic: per-sequence slabs folded in sequence order (no atomics) -> a second run is bitwise identical
    save2 = torch.empty_like(save)
    y2 = torch.empty_like(y_h)
    hip.ff_decoder_fwd(P.to(DEV), pe.to(DEV), cross.to(DEV), style.to(DEV), S, T, period, y2, save2)
    dP2, dc2, ds2 = torch.empty(nP, device=DEV), torch.empty(S, T, 64, device=DEV), torch.empty(S, 64, device=DEV)
    hip.ff_decoder_bwd(P.to(DEV), dy.to(DEV), S, T, period, save2, dP2, dc2, ds2)
    assert torch.equal(dP2, dP_h) and torch.equal(dc2, dc_h) and torch.equal(ds2, ds_h)
    assert rel_l2(dc_h.cpu(), dc_e) < 1e-4
    assert rel_l2(ds_h.cpu(), ds_e) < 1e-4
    names = [("Wqkv", 192 * 64), ("bqkv", 192), ("Wo", 4096), ("bo", 64), ("W1", 8192), ("b1", 128), ("W2", 8192), ("b2", 64),
             ("ln", 384), ("A", 4096), ("c", 64)]
    o = 0
    for nm, sz_ in names:
        assert rel_l2(dP_h.cpu()[o:o + sz_], dP_e[o:o + sz_]) < 1e-4, nm
        o += sz_


@pytest.mark.parametrize("S,T,period", [(1, 5, 60), (3, 70, 60), (2, 133, 7), (2, 290, 60)])
def test_ff_decoder_dropout_kernels_vs_spec(S, T, period):
    """a2f_ff_decoder_fwd_drop / bwd_drop (one keep mask per frame row at the PPE output, the self-attention probabilities, dropout1, the
    feed-forward activation and dropout3) against autograd on the float64 statement of the same masked recurrence; masks come from
    a2f_keep_bits as the model draws them; a second run is bitwise identical."""
    nP = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64 + 6 * 64 + 64 * 64 + 64
    P = rnd(nP, seed=70, scale=0.12)
    o_ln = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64
    P[o_ln:o_ln + 64] += 1.0; P[o_ln + 128:o_ln + 192] += 1.0; P[o_ln + 256:o_ln + 320] += 1.0
    P[o_ln + 384:o_ln + 384 + 64 * 64] *= 0.3
    pe = rnd(period, 64, seed=71)
    cross, style, dy = rnd(S, T, 64, seed=72), rnd(S, 64, seed=73), rnd(S, T, 64, seed=74)
    hip, emu = ops.HipBackend(), EmuBackend()
    row = torch.empty(S * T, 10, dtype=torch.int32, device=DEV)
    attn = torch.empty(S * 4 * T, hip.attn_keep_words(T), dtype=torch.int32, device=DEV)
    hip.keep_bits(4242, 1000, S * T, 320, 0.1, row)
    hip.keep_bits(4242, 1001, S * 4 * T, T, 0.1, attn)
    drop_h, drop_e = (row, attn, 1 / 0.9), (row.cpu(), attn.cpu(), 1 / 0.9)
    outs = []
    for _rep in range(2):
        y_h = torch.empty(S, T, 64, device=DEV)
        save = torch.empty(hip.ff_save_floats(S, T), device=DEV)
        hip.ff_decoder_fwd(P.to(DEV), pe.to(DEV), cross.to(DEV), style.to(DEV), S, T, period, y_h, save, drop=drop_h)
        dP_h, dc_h, ds_h = torch.full((nP,), 7.0, device=DEV), torch.empty(S, T, 64, device=DEV), torch.empty(S, 64, device=DEV)
        hip.ff_decoder_bwd(P.to(DEV), dy.to(DEV), S, T, period, save, dP_h, dc_h, ds_h, drop=drop_h)
        outs.append((y_h, dP_h, dc_h, ds_h))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    key = ("drop", S, T, period)
    if key not in _FF_SPEC:
        y_e = torch.empty(S, T, 64)
        emu.ff_decoder_fwd(P, pe, cross, style, S, T, period, y_e, None, drop=drop_e)
        dP_e, dc_e, ds_e = torch.zeros(nP), torch.empty(S, T, 64), torch.empty(S, 64)
        emu.ff_decoder_bwd(P, dy, S, T, period, None, dP_e, dc_e, ds_e, drop=drop_e)
        y_plain = torch.empty(S, T, 64)
        emu.ff_decoder_fwd(P, pe, cross, style, S, T, period, y_plain, None)
        _FF_SPEC[key] = (y_e, dP_e, dc_e, ds_e, y_plain)
    y_e, dP_e, dc_e, ds_e, y_plain = _FF_SPEC[key]
    y_h, dP_h, dc_h, ds_h = outs[0]
    assert rel_l2(y_h.cpu(), y_e) < 2e-5
    assert rel_l2(y_h.cpu(), y_plain) > 1e-2                      # the masks do something
    assert rel_l2(dc_h.cpu(), dc_e) < 1e-4 and rel_l2(ds_h.cpu(), ds_e) < 1e-4
    o = 0
    for nm, sz_ in [("Wqkv", 192 * 64), ("bqkv", 192), ("Wo", 4096), ("bo", 64), ("W1", 8192), ("b1", 128), ("W2", 8192), ("b2", 64), ("ln", 384),
                    ("A", 4096), ("c", 64)]:
        assert rel_l2(dP_h.cpu()[o:o + sz_], dP_e[o:o + sz_]) < 1e-4, nm
        o += sz_


def test_ff_decoder_bitwise_stable_under_contention():
    """Forward + blocked backward (five blocks: cross-block dq kernel, per-block recurrence and dK / dV kernels, weight-gradient products) on a
    high-priority stream while the library's GEMM kernels keep the chip busy on another stream, against the same call on an idle chip: every output
    bitwise equal, the save buffer filled with a different constant each time.  (A variant of the cross-block dq kernel that staged keys through
    LDS passed every other test and failed exactly this: its results changed when other kernels competed for the CUs.)"""
    from a2f_amd.ops import MMOp
    S, T, period = 24, 600, 60
    be = ops.HipBackend()
    nP = 192 * 64 + 192 + 64 * 64 + 64 + 128 * 64 + 128 + 64 * 128 + 64 + 6 * 64 + 64 * 64 + 64
    P = rnd(nP, seed=70, scale=0.1).to(DEV)
    pe = rnd(period, 64, seed=71).to(DEV)
    cross, style, dy = rnd(S, T, 64, seed=72).to(DEV), rnd(S, 64, seed=73).to(DEV), rnd(S, T, 64, seed=74).to(DEV)
    row = torch.empty(S * T, 10, dtype=torch.int32, device=DEV)
    attn = torch.empty(S * 4 * T, be.attn_keep_words(T), dtype=torch.int32, device=DEV)
    be.keep_bits(4242, 1000, S * T, 320, 0.1, row)
    be.keep_bits(4242, 1001, S * 4 * T, T, 0.1, attn)
    drop = (row, attn, 1 / 0.9)
    side = torch.cuda.Stream(priority=-1)
    a, w, o = rnd(9600, 768, seed=1).to(DEV), rnd(3072, 768, seed=2).to(DEV), torch.empty(9600, 3072, device=DEV)
    outs = []
    for rep in range(4):
        y = torch.empty(S, T, 64, device=DEV)
        save = torch.full((be.ff_save_floats(S, T),), float(rep), device=DEV)
        dP, dc, ds = torch.empty(nP, device=DEV), torch.empty(S, T, 64, device=DEV), torch.empty(S, 64, device=DEV)
        torch.cuda.synchronize()
        if rep % 2:
            for _ in range(300):
                be.mm(MMOp(a, 768), MMOp(w, 768), o, 9600, 3072, 768, 3072)
        with torch.cuda.stream(side):
            be.ff_decoder_fwd(P, pe, cross, style, S, T, period, y, save, drop=drop)
            be.ff_decoder_bwd(P, dy, S, T, period, save, dP, dc, ds, drop=drop)
        torch.cuda.synchronize()
        outs.append((y, dP, dc, ds))
    for rep in range(1, 4):
        assert all(torch.equal(u, v) for u, v in zip(outs[0], outs[rep])), rep


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K", [(32, 111, 5000), (32, 96, 777), (64, 96, 3001), (64, 192, 1200), (72, 72, 640), (128, 72, 19), (50, 128, 2048), (1, 1, 3)])
def test_wgrad_kernel_vs_contract(M, N, K, prec):
    """dW = dY^T X with the ones column and a conv padding mask, against the float64 emulation of the contract; prec 0 = exact
    fp32 MFMA, 1 = bf16x3 split products (16-row register sets, 96-column sub-problems): both inside 2e-5."""
    dy, x = rnd(K, M, seed=80), rnd(K + 2, N, seed=81)
    period = 4
    hip, emu = ops.HipBackend(), EmuBackend()
    hip.wgrad_precision = prec
    assert hip.wgrad_supported(M, N)
    outs = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        Cc = torch.full((M, N + 2), float("nan"), device=dev)
        db = torch.full((M,), float("nan"), device=dev)
        # B is a window view starting one element before row 0 (offset -1 masked by the padding rule on phase-0 rows)
        be.wgrad(dy.to(dev), x.to(dev), Cc, M, N, K, (0, 1, M), (0, N, 1), N + 2, ones_out=db, mask=(2, period, 0, 0, min(5, N)))
        outs.append((Cc.cpu(), db.cpu()))
    (c_h, d_h), (c_e, d_e) = outs
    assert rel_l2(c_h[:, :N], c_e[:, :N]) < 2e-5 and rel_l2(d_h, d_e) < 2e-5
    assert bool(torch.isnan(c_h[:, N:]).all())               # columns beyond N untouched
    # determinism (fixed fold order)
    Cc2 = torch.empty(M, N + 2, device=DEV); db2 = torch.empty(M, device=DEV)
    hip.wgrad(dy.to(DEV), x.to(DEV), Cc2, M, N, K, (0, 1, M), (0, N, 1), N + 2, ones_out=db2, mask=(2, period, 0, 0, min(5, N)))
    assert torch.equal(Cc2[:, :N].cpu(), c_h[:, :N])


def test_wgrad_batch_conv_layout_and_linears():
    """One launch pair for several problems: conv weight gradients stored straight in nn.Conv2d layout (c_unpack_kh) next
    to Linear weight gradients, against torch autograd; frame counts that leave ragged K tails."""
    be = ops.HipBackend()
    items, checks = [], []
    for li, (B_, P_out, Cin, Co) in enumerate([(37, 8, 37, 32), (21, 4, 32, 32), (13, 2, 32, 64), (45, 1, 64, 64)]):
        act = rnd(B_, 2 * P_out, Cin, seed=300 + li)
        dy = rnd(B_, P_out, Co, seed=310 + li)
        x = act.permute(0, 2, 1).clone()
        w = torch.zeros(Co, Cin, 3, requires_grad=True)
        bias = torch.zeros(Co, requires_grad=True)
        torch.nn.functional.conv1d(x, w, bias, stride=2, padding=1).backward(dy.permute(0, 2, 1))
        dw = torch.full((Co, Cin, 3, 1), float("nan"), device=DEV)
        db = torch.full((Co,), float("nan"), device=DEV)
        keep = (act.to(DEV), dy.to(DEV))
        items.append(ops.conv_wgrad_item(be, keep[0], keep[1], dw, db, B_ * P_out, P_out, Cin, Co))
        checks.append((dw, w.grad.unsqueeze(-1), db, bias.grad, keep))
    for li, (M, N, K) in enumerate([(333, 72, 72), (65, 128, 72), (1001, 50, 128)]):
        dy, xx = rnd(M, N, seed=320 + li), rnd(M, K, seed=330 + li)
        dW = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        keep = (dy.to(DEV), xx.to(DEV))
        items.append(ops.linear_wgrad_item(be, keep[0], keep[1], dW, db, M, N, K))
        checks.append((dW, dy.double().t() @ xx.double(), db, dy.double().sum(0), keep))
    be.wgrad_batch(items)
    for got_w, ref_w, got_b, ref_b, _ in checks:
        assert rel_l2(got_w.cpu(), ref_w) < 2e-5 and rel_l2(got_b.cpu(), ref_b) < 2e-5
    with pytest.raises(a2f_amd.A2FError):                           # more than one batch worth in a single C call
        arr = (type(items[0][0]) * 9)(*[it[0] for it in (items + items)[:9]])
        from a2f_amd import _lib
        _lib.check(be.lib.a2f_wgrad_batch_f32(arr, 9, None), "a2f_wgrad_batch_f32")


@pytest.mark.parametrize("M,N,K", [(512, 32, 111), (777, 64, 96), (1500, 128, 72), (640, 50, 128), (3000, 64, 192), (513, 72, 72), (900, 37, 33)])
@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_gemm_tall_skinny_shapes(M, N, K, layout):
    """The trunk's tall-skinny products (M >= 512, K <= 192, N <= 128): all epilogues, ldc > N, accumulate."""
    A = rnd(M, K, seed=90)
    if layout == "nt":
        B, b = rnd(N, K, seed=91), (0, 1, K)
    else:
        B, b = rnd(K, N + 3, seed=91), (0, N + 3, 1)
    bias, aux, C0 = rnd(N, seed=92), rnd(M, N, seed=93), rnd(M, N + 2, seed=94)
    for kw in (dict(), dict(bias=bias, act=1), dict(act=2), dict(aux=aux, ldaux=N, aux_mode=1), dict(aux=aux, ldaux=N, aux_mode=2, accumulate=True),
               dict(bias=bias, aux=aux, ldaux=N, aux_mode=3)):
        (c_hip, _), (c_emu, _) = run_gemm_both(A, B, C0, M, N, K, (0, K, 1), b, N + 2, **kw)
        assert rel_l2(c_hip, c_emu) < 2e-5, kw.keys()
        assert torch.equal(c_hip[:, N:], C0[:, N:])


def test_conv_window_masks_large():
    """Implicit-GEMM conv forward / shifted dgrad windows at M = 1600 rows against torch conv1d."""
    B_, P_out, Cin, Co = 200, 8, 37, 32
    P_in = 2 * P_out
    act = rnd(B_, P_in, Cin, seed=95)
    w = rnd(Co, Cin, 3, seed=96, scale=0.2)
    bias = rnd(Co, seed=97)
    x = act.permute(0, 2, 1).clone().requires_grad_(True)
    y = torch.relu(torch.nn.functional.conv1d(x, w, bias, stride=2, padding=1))
    dy = rnd(B_, P_out, Co, seed=98)
    y.backward(dy.permute(0, 2, 1))
    be = ops.HipBackend()
    wg = torch.empty(Co, 3 * Cin, device=DEV)
    be.conv_weight_pack(w.to(DEV), wg, Co, Cin, 3, 0)
    out = torch.empty(B_, P_out, Co, device=DEV)
    R = B_ * P_out
    ops.conv_fwd(be, act.to(DEV), wg, bias.to(DEV), out, R, P_out, Cin, Co)
    assert rel_l2(out.cpu(), y.detach().permute(0, 2, 1)) < 2e-5
    d_pre = (dy * (out.cpu() > 0)).to(DEV).contiguous()
    d_in = torch.full((B_, P_in, Cin), float("nan"), device=DEV)
    ops.conv_k3s2_bwd_input(be, d_pre, wg, d_in, R, P_out, Cin, Co)
    assert rel_l2(d_in.cpu(), x.grad.permute(0, 2, 1)) < 5e-5


@pytest.mark.parametrize("N,Lin,Lout,C_", [(2, 499, 600, 512), (1, 2499, 3000, 512), (3, 600, 499, 64), (2, 1, 7, 8), (2, 9, 1, 8), (1, 50, 50, 4)])
def test_interp_linear_matches_torch_interpolate(N, Lin, Lout, C_):
    """csrc/interp.hip against the torch fp32 CPU op it replaces (wav2vec.py:76-84: F.interpolate on the transposed view,
    align_corners=True), forward and backward; tolerance 1e-6 relative L2 (fp32, fused multiply-add differences only)."""
    import torch.nn.functional as F
    from a2f_amd.functional import LinearInterpFn
    x = rnd(N, Lin, C_, seed=70).to(DEV).requires_grad_(True)
    gy = rnd(N, Lout, C_, seed=71).to(DEV)
    y = LinearInterpFn.apply(x, Lout)
    y.backward(gy)
    # the reference path is the CPU op (torch's HIP kernel fuses the index arithmetic and lands an ulp of the source
    # position away: 1e-5 relative at L = 600)
    x2 = x.detach().cpu().clone().requires_grad_(True)
    y2 = F.interpolate(x2.transpose(1, 2), size=Lout, align_corners=True, mode="linear").transpose(1, 2)
    y2.backward(gy.cpu())
    assert y.shape == y2.shape
    assert rel_l2(y.detach().cpu(), y2.detach().cpu()) < 1e-6
    assert rel_l2(x.grad.cpu(), x2.grad.cpu()) < 1e-6


def test_cabi_from_a_plain_host_program():
    """The boundary is a C ABI, not a torch extension: tools/cabi_smoke.cpp (no Python, no torch; device buffers from the HIP
    runtime) calls a2f_head_fused and a2f_vocaloss_fwd and checks them against a double-precision host evaluation of the
    reference formulas (voca.py:48-49, loss.py:24-55)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as ge
    exe = ge.build_cabi_smoke()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CABI_SMOKE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_head_fused_at_bench_size_properties():
    """BASELINE configs[1] size (19 200 frames x 15 069 columns, K = 50), where the CPU contract would take minutes: size
    -independent properties instead.  (a) pred of sampled rows against a float64 evaluation; (b) the batch split into two even
    halves gives the same pred rows bit for bit, the mean of the two losses, half-scaled dH rows and the mean of the dW / db;
    (c) the standalone VocaLoss kernel agrees with the fused loss."""
    B, n, K = 19200, 15069, 50
    hip = ops.HipBackend()
    g = torch.Generator(device=DEV).manual_seed(77)
    H = torch.randn(B, K, device=DEV, generator=g); Wt = torch.randn(n, K, device=DEV, generator=g) * 0.3
    bias = torch.randn(n, device=DEV, generator=g)
    templ = torch.randn(B, n, device=DEV, generator=g) * 5; gt = templ + torch.randn(B, n, device=DEV, generator=g) * 0.5

    def run(lo, hi):
        b = hi - lo
        pred = torch.empty(b, n, device=DEV); out3 = torch.empty(3, device=DEV)
        dH, dW, db = torch.empty(b, K, device=DEV), torch.empty(n, K, device=DEV), torch.empty(n, device=DEV)
        hip.head_fused(H[lo:hi].contiguous(), Wt, bias, templ[lo:hi].contiguous(), b, gt[lo:hi].contiguous(), b, b, n, K, 1.0, 10.0, pred, out3, dH, dW, db)
        return pred, out3, dH, dW, db
    p_f, o_f, dH_f, dW_f, db_f = run(0, B)
    rows = torch.randint(0, B, (48,), generator=torch.Generator().manual_seed(3)).to(DEV)
    ref = H[rows].double() @ Wt.double().t() + bias.double() + templ[rows].double()
    assert rel_l2((p_f[rows].double() - templ[rows].double()).cpu(), (ref - templ[rows].double()).cpu()) < 2e-5
    p_a, o_a, dH_a, dW_a, db_a = run(0, B // 2)
    p_b, o_b, dH_b, dW_b, db_b = run(B // 2, B)
    assert torch.equal(p_f[:B // 2], p_a) and torch.equal(p_f[B // 2:], p_b)
    assert torch.allclose(o_f, (o_a + o_b) / 2, rtol=1e-5, atol=0)
    assert rel_l2(dH_f.cpu(), torch.cat([dH_a, dH_b]).cpu() / 2) < 1e-5
    assert rel_l2(dW_f.cpu(), ((dW_a + dW_b) / 2).cpu()) < 1e-5 and rel_l2(db_f.cpu(), ((db_a + db_b) / 2).cpu()) < 1e-5
    l3 = torch.empty(3, device=DEV)
    hip.vocaloss_fwd(p_f, gt, B, n, 1.0, 10.0, l3)
    assert torch.allclose(l3, o_f, rtol=1e-5, atol=0)


@pytest.mark.parametrize("training,relu", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("groups,P_out,Cc", [(24, 16, 72), (5, 3, 8), (300, 16, 72)])
def test_bn_conv0_fused_vs_contract(groups, P_out, Cc, training, relu):
    """BatchNorm with the one-input-channel convolution recomputed in every pass (a2f_bn_fwd_conv0 / a2f_bn_bwd_conv0, Audio2Mesh analysis layer 0)
    against the float64 emulation of convolution + BatchNorm: output, saved statistics, running buffers, dx, dgamma, dbeta."""
    hip, emu = ops.HipBackend(), EmuBackend()
    R = groups * P_out
    a0, wg, cb = rnd(groups, 2 * P_out, seed=80), rnd(Cc, 3, seed=81), rnd(Cc, seed=82)
    gamma, beta = rnd(Cc, seed=83) + 1.5, rnd(Cc, seed=84) * 0.3
    dy = rnd(R, Cc, seed=85)
    res = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        rm, rv = torch.zeros(Cc, device=dev) + 0.1, torch.ones(Cc, device=dev) * 0.9
        y, mu, istd = torch.empty(R, Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        args = [t.to(dev) for t in (a0, wg, cb)]
        be.bn_fwd_conv0(*args, P_out, R, Cc, gamma.to(dev), beta.to(dev), rm, rv, 0.1, 1e-5, training, relu, y, mu, istd)
        dx, dg, db = torch.empty(R, Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        be.bn_bwd_conv0(*args, P_out, beta.to(dev), dy.to(dev), R, Cc, gamma.to(dev), mu, istd, training, relu, dx, dg, db)
        # the same pass with the convolution's own weight / bias gradient collected in it: once beside dx, once with dx never written (the Audio2Mesh step's form)
        dx2, dg2, db2, dwg, dcb = torch.empty(R, Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, 3, device=dev), torch.empty(Cc, device=dev)
        be.bn_bwd_conv0(*args, P_out, beta.to(dev), dy.to(dev), R, Cc, gamma.to(dev), mu, istd, training, relu, dx2, dg2, db2, dwg=dwg, dcbias=dcb)
        dg3, db3, dwg3, dcb3 = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, 3, device=dev), torch.empty(Cc, device=dev)
        be.bn_bwd_conv0(*args, P_out, beta.to(dev), dy.to(dev), R, Cc, gamma.to(dev), mu, istd, training, relu, None, dg3, db3, dwg=dwg3, dcbias=dcb3)
        if be is hip:
            assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
            assert torch.equal(dwg3, dwg) and torch.equal(dcb3, dcb) and torch.equal(dg3, dg) and torch.equal(db3, db)
        res.append([t.cpu() for t in (y, mu, istd, rm, rv, dx, dg, db, dwg, dcb)])
    for name, h, e in zip(("y", "mean", "invstd", "running_mean", "running_var", "dx", "dgamma", "dbeta", "conv0 dW", "conv0 dbias"), *res):
        if name == "conv0 dbias":
            # under batch statistics the columns of dx sum to zero identically (a bias in front of a BatchNorm has no gradient): what is left is rounding, so it
            # is measured against the size of the terms of the sum, not against the sum
            scale = res[1][5].double().abs().sum(dim=0).norm()
            assert float((h.double() - e.double()).norm() / scale) < 2e-6, name
            continue
        assert rel_l2(h, e) < 2e-5, name


def test_relayout_batch_vs_contract():
    """a2f_relayout_batch: packed (mode 0), transposed-packed (mode 1) weights with padded extents and pad values, gradient unpack (mode 2), vectors (Ci = KH = 1);
    40 items, i.e. two launches; every destination element bit-equal to the emulation (pure data movement)."""
    hip, emu = ops.HipBackend(), EmuBackend()
    shapes = [(72, 1, 3, 72, 1), (108, 72, 3, 108, 72), (162, 108, 3, 164, 108), (243, 162, 3, 244, 164), (256, 243, 3, 256, 244), (256, 256, 4, 256, 256), (162, 1, 1, 164, 1)]
    res = []
    for be, dev in ((hip, DEV), (emu, "cpu")):
        items, outs = [], []
        for rep in range(2):
            for i, (Co, Ci, KH, Cop, Cip) in enumerate(shapes):
                w = rnd(Co, Ci, KH, seed=900 + i + 10 * rep).to(dev)
                for mode in (0, 1):
                    dst = torch.full((Cop * KH * Cip,), float("nan"), device=dev)
                    items.append((w, dst, Co, Ci, KH, Cop, Cip, mode, 1.0 if Ci == 1 and KH == 1 else 0.0))
                    outs.append(dst)
                g = rnd(Cop, KH * Cip, seed=950 + i + 10 * rep).to(dev)
                dst = torch.full((Co * Ci * KH,), float("nan"), device=dev)
                items.append((g, dst, Co, Ci, KH, Cop, Cip, 2))
                outs.append(dst)
        assert len(items) > 32
        be.relayout_batch(items)
        res.append([o.cpu() for o in outs])
    for h, e in zip(*res):
        assert torch.equal(h, e)
    with pytest.raises(a2f_amd.A2FError):
        hip.relayout_batch([(torch.zeros(8, device=DEV), torch.zeros(8, device=DEV), 4, 4, 1, 4, 4, 0)])      # 16 source elements claimed, 8 given
