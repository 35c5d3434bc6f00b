"""GPU parity of the drop-in modules (through the C ABI) against the reference goldens and the CPU oracle.

Bar (BASELINE.json north_star): <= 1e-4 relative L2 on fp32 vertex offsets; gradients held to the same bar.
"""
import os

import pytest
import torch

import a2f_amd
from golden_util import load, packed_error, rel_l2
from a2f_amd import ops
from oracle import losses as o_loss
from oracle import voca as o_voca
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda"
NV = 15069
TOL = 1e-4


def _voca(dev=DEV):
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    sd = W.formula_state_dict(W.voca_shapes(NV), gains={"decoder.4": 4.0})
    m.load_state_dict(sd)
    return m.to(dev), sd


def test_voca_golden_g1():
    g = load("g1_voca")
    m, _ = _voca()
    B = 4
    x = W.formula_input("g1.x", (B, 29, 16)).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g1.template", (B, 5023, 3), 5.0)
    gt = (template + W.formula_input("g1.gt", (B, 5023, 3), 0.5)).to(DEV)
    template = template.to(DEV)
    pred = m(x, one_hot, template)
    assert pred.shape == (B, 5023, 3) and pred.is_cuda
    assert packed_error(g, "pred", pred) < TOL
    ref_off = torch.from_numpy(g["pred"]) - template.cpu()
    assert rel_l2(pred.cpu() - template.cpu(), ref_off) < TOL                 # offsets alone
    loss = a2f_amd.VocaLoss()(pred, gt)
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g["loss::" + k])) <= 1e-5 * abs(float(g["loss::" + k]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        assert packed_error(g, "grad::" + k, p.grad) < TOL, k


@pytest.mark.parametrize("B", [2, 130, 1024])
def test_voca_vs_oracle_seeded(B):
    """Same seeded inputs through the HIP path and the CPU oracle (sizes the oracle finishes in seconds)."""
    m, sd = _voca()
    gen = torch.Generator().manual_seed(1234 + B)
    x = torch.randn(B, 29, 16, generator=gen)
    one_hot = torch.zeros(B, 12)
    one_hot[torch.arange(B), torch.randint(0, 12, (B,), generator=gen)] = 1.0
    template = torch.randn(B, 5023, 3, generator=gen) * 5
    gt = template + torch.randn(B, 5023, 3, generator=gen) * 0.5
    pred = m(x.to(DEV), one_hot.to(DEV), template.to(DEV))
    loss = a2f_amd.VocaLoss()(pred, gt.to(DEV))
    loss["loss"].backward()
    # float64 oracle = the truth both fp32 paths approximate (a ReLU unit whose pre-activation rounds across zero
    # flips a whole gradient path: the fp32 CPU oracle itself is 4e-3 away from fp64 on time_conv.0.weight at B=1024)
    P = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    op = o_voca.voca_forward(P, x.double(), one_hot.double(), template.double())
    ol = o_loss.voca_loss(op, gt.double())
    ol["loss"].backward()
    assert rel_l2(pred.cpu() - template, op.detach() - template) < TOL
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(ol[k].detach())) <= 2e-5 * abs(float(ol[k].detach()))
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu(), P[k].grad) < TOL, k


def test_voca_predict_and_eval_no_grad():
    m, _ = _voca()
    x = torch.randn(6, 29, 16, device=DEV)
    oh = torch.zeros(6, 12, device=DEV); oh[:, 1] = 1
    t = torch.randn(6, 5023, 3, device=DEV)
    with torch.no_grad():
        a = m.predict(x, oh, t)
        b = m(x, oh, t, extra_kwarg=1)
    assert torch.equal(a, b)


def test_cpu_tensors_fail_loudly():
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    with pytest.raises(a2f_amd.A2FError):
        m(torch.zeros(2, 29, 16), torch.zeros(2, 12), torch.zeros(2, 5023, 3))


def test_voca_fused_step_golden_g1():
    g = load("g1_voca")
    m, _ = _voca()
    B = 4
    x = W.formula_input("g1.x", (B, 29, 16)).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g1.template", (B, 5023, 3), 5.0)
    gt = (template + W.formula_input("g1.gt", (B, 5023, 3), 0.5)).to(DEV)
    pred, loss = m.forward_with_loss(x, one_hot, template.to(DEV), gt)
    assert packed_error(g, "pred", pred) < TOL
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g["loss::" + k])) <= 1e-5 * abs(float(g["loss::" + k]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        assert packed_error(g, "grad::" + k, p.grad) < TOL, k


def test_voca_fused_equals_unfused_large():
    """Fused single-pass head+loss+backward vs the separate kernels at a bench-like shape (B = 2 x 600 frames)."""
    torch.manual_seed(0)
    m1, _ = _voca()
    m2, _ = _voca()
    B = 1200
    x = torch.randn(B, 29, 16, device=DEV)
    oh = torch.zeros(B, 12, device=DEV); oh[torch.arange(B), torch.arange(B) % 8] = 1
    t = torch.randn(B, 5023, 3, device=DEV) * 5
    gt = t + torch.randn(B, 5023, 3, device=DEV) * 0.5
    l1 = a2f_amd.VocaLoss()(m1(x, oh, t), gt)
    l1["loss"].backward()
    p2, l2 = m2.forward_with_loss(x, oh, t, gt)
    l2["loss"].backward()
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(l1[k].detach()) - float(l2[k].detach())) <= 2e-5 * abs(float(l1[k].detach()))
    for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert rel_l2(b.grad, a.grad) < 5e-5, k


def test_voca_two_phase_graph_step_equals_single_graph():
    """The data-parallel overlap form (two captured graphs split after the fused head, segment-wise all-reduce hooks)
    must leave exactly the parameters of the single-graph step (world size 1: the collectives are no-ops)."""
    from a2f_amd.trainer import TrainStep
    flats = []
    for overlap in (False, True):
        torch.manual_seed(7)
        m = a2f_amd.Voca(n_verts=NV, n_onehot=12).to(DEV)
        step = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-3, use_graph=True, overlap=overlap)
        assert step.overlap == overlap
        g = torch.Generator().manual_seed(3)
        B = 64
        x = torch.randn(B, 29, 16, generator=g).to(DEV)
        oh = torch.zeros(B, 12); oh[torch.arange(B), torch.arange(B) % 8] = 1
        t = torch.randn(B, 5023, 3, generator=g) * 5
        gt = (t + 0.5 * torch.randn(B, 5023, 3, generator=g)).to(DEV)
        oh, t = oh.to(DEV), t.to(DEV)
        for _ in range(5):                                 # 2 eager warm-ups, capture, 2 replays
            out = step(x, oh, gt, t)
        torch.cuda.synchronize()
        flats.append((step.opt.flat_p.clone(), float(out["loss"])))
    assert torch.equal(flats[0][0], flats[1][0]) and flats[0][1] == flats[1][1]


BN_CANCELLED = {f"analysis_net.{3 * l}.bias" for l in range(5)} | {f"articulation_net.{i}.bias" for i in (0, 3, 6)}


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_audio2mesh_golden_g2(mode):
    """Convolutions as windowed a2f_mm products in bf16x6; forward, losses, EVERY gradient <= 1e-4, BN running statistics."""
    g = load("g2_audio2mesh")
    m = a2f_amd.Audio2Mesh(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.audio2mesh_shapes(NV, 12), gains={"output_net.4": 4.0}))
    m.to(DEV).train(mode == "train")
    B = 4
    x = W.formula_input("g2.x", (B, 52, 32)).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g2.template", (B, 5023, 3), 5.0)
    gt = (template + W.formula_input("g2.gt", (B, 5023, 3), 0.5)).to(DEV)
    pred = m(x, one_hot, template.to(DEV))
    assert packed_error(g, f"{mode}::pred", pred) < TOL
    loss = a2f_amd.VocaLoss()(pred, gt)
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g[f"{mode}::loss::{k}"])) <= 2e-5 * abs(float(g[f"{mode}::loss::{k}"]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        if mode == "train" and k in BN_CANCELLED:
            # exact gradient is 0 (batch-stat BN removes the conv bias); what is left is fp32 cancellation noise
            wgrad = dict(m.named_parameters())[k.replace(".bias", ".weight")].grad
            assert float(p.grad.norm()) < 1e-3 * float(wgrad.norm()), k
            continue
        assert packed_error(g, f"{mode}::grad::{k}", p.grad) < TOL, k
    if mode == "train":
        for k, b in m.named_buffers():
            if b.is_floating_point():
                assert packed_error(g, "train::buffer::" + k, b) < 1e-5, k


def test_audio2mesh_fused_training_path_golden_g2():
    """forward_with_loss (output_net.4 + template + VocaLoss + head backward in one pass) against the same G2 fixture:
    prediction, loss terms and every gradient (train mode: BN batch statistics)."""
    g = load("g2_audio2mesh")
    m = a2f_amd.Audio2Mesh(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.audio2mesh_shapes(NV, 12), gains={"output_net.4": 4.0}))
    m.to(DEV).train()
    B = 4
    x = W.formula_input("g2.x", (B, 52, 32)).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g2.template", (B, 5023, 3), 5.0)
    gt = (template + W.formula_input("g2.gt", (B, 5023, 3), 0.5)).to(DEV)
    pred, loss = m.forward_with_loss(x, one_hot, template.to(DEV), gt)
    assert packed_error(g, "train::pred", pred) < TOL
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g[f"train::loss::{k}"])) <= 2e-5 * abs(float(g[f"train::loss::{k}"]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        if k in BN_CANCELLED:
            wgrad = dict(m.named_parameters())[k.replace(".bias", ".weight")].grad
            assert float(p.grad.norm()) < 1e-3 * float(wgrad.norm()), k
            continue
        assert packed_error(g, f"train::grad::{k}", p.grad) < TOL, k


def test_audio2mesh_audio_sample_windows_g7():
    """Config 1 plumbing: windows cut from assets/audio_sample.npy, batch = 1 per frame, eval mode."""
    g = load("g7_audio_sample_audio2mesh")
    m = a2f_amd.Audio2Mesh(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.audio2mesh_shapes(NV, 12), gains={"output_net.4": 4.0}))
    m.to(DEV).eval()
    x = torch.from_numpy(g["x"]).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g7.template", (1, 5023, 3), 5.0).to(DEV)
    with torch.no_grad():
        pred = torch.cat([m(x[i:i + 1], one_hot, template) for i in range(x.shape[0])], 0)
    assert packed_error(g, "pred", pred) < TOL


FF_GAINS = {"vertice_map_r.weight": 4.0, "vertice_map.weight": 0.05, "obj_vector": 2.0, "in_proj_weight": 1.5}


def _faceformer(stub=True):
    m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12, audio_encoder=torch.nn.Identity() if stub else None)
    m.load_state_dict(W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS), strict=False)
    return m.to(DEV).eval()          # the deterministic function; the train-mode tests call .train() and inject / seed their masks


@pytest.mark.parametrize("T", [7, 61, 130])
def test_faceformer_decoder_golden_g3(T):
    """Decoder + head + FaceFormerLoss with a pinned encoder output (crosses the period-60 boundary, odd-T branch)."""
    g = load("g3_faceformer_decoder")
    m = _faceformer()
    pre = f"T{T}::"
    enc = W.formula_input(f"g3.enc{T}", (1, T, 768)).to(DEV).requires_grad_(True)
    audio = torch.zeros(1, T * 16000 // 60 + 8, device=DEV)
    one_hot = torch.from_numpy(g[pre + "one_hot"]).to(DEV)
    template = W.formula_input(f"g3.template{T}", (1, 5023, 3), 5.0)
    gt = (template.view(1, 1, 5023, 3) + W.formula_input(f"g3.gt{T}", (1, T, 5023, 3), 0.5)).to(DEV)
    for fused in (False, True):
        m.zero_grad()
        enc.grad = None
        if fused:
            pred, loss = m.forward_with_loss(audio, one_hot, template.to(DEV), gt, encoder_states=enc)
        else:
            pred = m(audio, one_hot, template.to(DEV), encoder_states=enc)
            loss = a2f_amd.FaceFormerLoss()(pred, gt)
        assert packed_error(g, pre + "pred", pred) < TOL
        import numpy as np
        assert np.allclose(pred.detach().double().reshape(T, -1).norm(dim=1).cpu().numpy(), g[pre + "pred::frame_l2"], rtol=1e-5)
        for k in ("loss", "rec_loss", "vel_loss"):
            assert abs(float(loss[k].detach()) - float(g[pre + "loss::" + k])) <= 1e-4 * abs(float(g[pre + "loss::" + k]))
        loss["loss"].backward()
        for k, p in m.named_parameters():
            if (pre + "nograd::" + k) in g:
                continue
            assert packed_error(g, pre + "grad::" + k, p.grad) < TOL, (fused, k)
        assert packed_error(g, pre + "grad::enc", enc.grad) < TOL


@pytest.mark.parametrize("images", [False, True])
def test_faceformer_end_to_end_golden_g8(images, monkeypatch):
    """images: the Transformer layers on split-32 image operands (the product's path from 2048 rows; forced here at T = 30)."""
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0 if images else 1 << 30)
    g = load("g8_faceformer_e2e")
    m = _faceformer(stub=False)
    m.audio_encoder.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0}))
    m.to(DEV).eval()
    T = 30
    audio = W.formula_input("g8.audio", (1, T * 16000 // 60 + 5), 0.1).to(DEV)
    one_hot = torch.from_numpy(g["one_hot"]).to(DEV)
    template = W.formula_input("g8.template", (1, 5023, 3), 5.0).to(DEV)
    with torch.no_grad():
        pred = m(audio, one_hot, template)
    assert packed_error(g, "pred", pred) < TOL


def test_faceformer_long_sequence_beyond_reference_limit():
    """T = 700 > 600: the reference's materialised mask/PPE tables stop at 600 frames; closed forms do not."""
    m = _faceformer()
    T = 700
    enc = torch.randn(1, T, 768, device=DEV)
    audio = torch.zeros(1, T * 16000 // 60 + 8, device=DEV)
    oh = torch.zeros(1, 12, device=DEV); oh[0, 3] = 1
    with torch.no_grad():
        pred = m(audio, oh, torch.zeros(1, 5023, 3, device=DEV), encoder_states=enc)
    assert pred.shape == (1, T, 5023, 3) and bool(torch.isfinite(pred).all())


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("B", [1, 31, 32, 33, 700])
def test_voca_fused_trunk_kernel_equals_layerwise(B, exact, monkeypatch):
    """csrc/voca_trunk.hip (one launch, activations in LDS) against the per-layer implicit-GEMM sequence (exact fp32 matrix
    instructions): every saved activation and the 50-wide bottleneck.  The fused kernel multiplies in split bf16 (three bf16 MFMAs per
    product, <= 2^-16 per product) by default and in exact fp32 under A2F_TRUNK_EXACT=1."""
    from a2f_amd import ops
    monkeypatch.setenv("A2F_TRUNK_EXACT", "1" if exact else "0")
    tol = 2e-6 if exact else 2e-5
    from a2f_amd.functional import voca_trunk_fwd, voca_trunk_fwd_layerwise
    m, _ = _voca()
    gen = torch.Generator().manual_seed(B)
    x = torch.randn(B, 29, 16, generator=gen).to(DEV)
    oh = torch.zeros(B, 12); oh[torch.arange(B), torch.randint(0, 12, (B,), generator=gen)] = 1.0
    oh = oh.to(DEV)
    params = [p.data for p in m._trunk_params()]
    be = ops.backend()
    h_f, s_f = voca_trunk_fwd(be, x, oh, params)
    h_l, s_l = voca_trunk_fwd_layerwise(be, x, oh, params)
    assert rel_l2(h_f, h_l) < tol
    for name, a, b in zip(("a0", "a1", "a2", "a3", "cat", "z1", "z2"), s_f[:7], s_l[:7]):
        assert rel_l2(a, b) < tol, name


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("B", [2, 33, 700])
def test_voca_fused_trunk_backward_equals_layerwise(B, exact, monkeypatch):
    """Fused data-gradient chain + wgrad contractions against the per-layer implicit-GEMM backward."""
    from a2f_amd import ops
    monkeypatch.setenv("A2F_TRUNK_EXACT", "1" if exact else "0")
    from a2f_amd.functional import voca_trunk_bwd, voca_trunk_bwd_layerwise, voca_trunk_fwd
    m, _ = _voca()
    gen = torch.Generator().manual_seed(100 + B)
    x = torch.randn(B, 29, 16, generator=gen).to(DEV)
    oh = torch.zeros(B, 12); oh[torch.arange(B), torch.randint(0, 12, (B,), generator=gen)] = 1.0
    oh = oh.to(DEV)
    dh3 = torch.randn(B, 50, generator=gen).to(DEV)
    be = ops.backend()
    _, saved = voca_trunk_fwd(be, x, oh, [p.data for p in m._trunk_params()])
    g_f = voca_trunk_bwd(be, saved, dh3)
    g_l = voca_trunk_bwd_layerwise(be, saved, dh3)
    for i, (a, b) in enumerate(zip(g_f, g_l)):
        assert rel_l2(a, b) < 2e-5, i


def test_faceformer_encoder_autocast_close_to_fp32():
    """`percision: 16-mixed`: the wav2vec2 encoder under bf16 autocast, decoder / head / loss in fp32 - same function to
    mixed-precision accuracy (the 1e-4 parity bar is defined on the fp32 setting)."""
    torch.manual_seed(0)
    m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12).to(DEV).eval()
    with torch.no_grad():
        m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
    T = 30
    audio = torch.randn(1, T * 16000 // 60 + 8, device=DEV) * 0.1
    oh = torch.zeros(1, 12, device=DEV); oh[0, 3] = 1
    t = torch.randn(1, 5023, 3, device=DEV)
    with torch.no_grad():
        ref = m(audio, oh, t)
        m.encoder_autocast = torch.bfloat16
        amp = m(audio, oh, t)
    assert amp.dtype == torch.float32 and rel_l2((amp - t.view(1, 1, -1, 3)).cpu(), (ref - t.view(1, 1, -1, 3)).cpu()) < 5e-2


def test_bench_two_rank_flow_rehearsal_on_one_gpu():
    """bench.py's multi-rank path end to end (torchrun launch, two-phase hipGraphs, overlapped segment all-reduce, max-over
    -ranks timing, rank-0-only roofline leg, barriers) with two ranks sharing cuda:0 and gloo collectives - RCCL needs one
    GPU per rank, which the test box does not have; everything else is the code the 2/4/8-GPU runs execute."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, A2F_BENCH_BACKEND="gloo", A2F_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3",
                        "--seqs", "4", "--frames", "64", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    assert "overlapped" in out["config"]["launch"] and out["roofline"]["kernel"] == "head_x3_kernel"


def test_bench_rccl_one_rank_rehearsal():
    """The RCCL side of the multi-GPU flow on the one GPU the test box has: a one-rank nccl (= RCCL) process group, the
    two-phase hipGraphs captured while RCCL's watchdog thread is alive, the asynchronous head-segment all-reduce issued
    between the two graph replays, the trunk-segment all-reduce, barriers, destroy_process_group."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, A2F_BENCH_FORCE_DIST="1", A2F_DP_FORCE_COLLECTIVES="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3",
                        "--seqs", "4", "--frames", "64", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "overlapped" in out["config"]["launch"]


@pytest.mark.parametrize("S,T", [(3, 40), (2, 601), (5, 32)])
def test_voca_trunk_on_the_fly_windows_equal_materialised_windows(S, T):
    """the trunk fed with per-frame features (S,T,29) builds the 16-frame zero-padded windows on chip;
    every output must be bit-identical to the same kernel fed with feature_windows(feats) (sequence boundaries fall inside
    32-frame blocks for these shapes)."""
    from a2f_amd import ops
    from a2f_amd.data import feature_windows
    from a2f_amd.functional import voca_trunk_fwd
    be = ops.backend()
    m = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
    params = [p.data for p in m._trunk_params()]
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(S, T, 29, generator=g)
    oh = torch.zeros(S * T, 12); oh[torch.arange(S * T), torch.arange(S * T) % 8] = 1
    h_w, s_w = voca_trunk_fwd(be, feature_windows(feats).to(DEV), oh.to(DEV), params)
    h_f, s_f = voca_trunk_fwd(be, feats.to(DEV), oh.to(DEV), params)
    assert torch.equal(h_w, h_f)
    for a, b in zip(s_w[:7], s_f[:7]):
        assert torch.equal(a, b)


def test_voca_train_step_from_features_matches_windows():
    from a2f_amd.data import feature_windows
    from a2f_amd.trainer import TrainStep
    S, T = 2, 48
    g = torch.Generator().manual_seed(6)
    feats = torch.randn(S, T, 29, generator=g)
    B = S * T
    oh = torch.zeros(B, 12); oh[:, 2] = 1
    templ = torch.randn(B, 5023, 3, generator=g) * 5
    gt = templ + torch.randn(B, 5023, 3, generator=g) * 0.5
    outs = []
    for x in (feature_windows(feats), feats):
        torch.manual_seed(11)
        m = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
        step = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-4)
        out = step(x.to(DEV), oh.to(DEV), gt.to(DEV), templ.to(DEV))
        outs.append((float(out["loss"]), [p.detach().clone() for p in m.parameters()]))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_faceformer_config5_long_context_prefix_property():
    """BASELINE configs[4] shape (T = 3000 frames, beyond the reference's 600-frame tables): the decoder is causal, so the
    first 600 frames of a 3000-frame sequence must equal the 600-frame run on the same encoder-state prefix (a size
    -independent parity property), and the fused loss / backward of the long sequence must be finite."""
    m = _faceformer()
    T, Tp = 3000, 600
    g = torch.Generator().manual_seed(12)
    enc = torch.randn(1, T, 768, generator=g).to(DEV)
    oh = torch.zeros(1, 12, device=DEV); oh[0, 5] = 1
    templ = (torch.randn(1, 5023, 3, generator=g) * 5).to(DEV)
    with torch.no_grad():
        long = m(torch.zeros(1, T * 16000 // 60 + 8, device=DEV), oh, templ, encoder_states=enc)
        short = m(torch.zeros(1, Tp * 16000 // 60 + 8, device=DEV), oh, templ, encoder_states=enc[:, :Tp].contiguous())
    assert long.shape == (1, T, 5023, 3) and bool(torch.isfinite(long).all())
    t4 = templ.view(1, 1, 5023, 3)
    assert rel_l2((long[:, :Tp] - t4).cpu(), (short - t4).cpu()) < 1e-5
    gt = long.detach() + 0.5 * torch.randn(long.shape, generator=g).to(DEV)
    enc_g = enc.clone().requires_grad_(True)
    pred = m(torch.zeros(1, T * 16000 // 60 + 8, device=DEV), oh, templ, encoder_states=enc_g)
    loss = a2f_amd.FaceFormerLoss()(pred, gt)
    loss["loss"].backward()
    assert bool(torch.isfinite(loss["loss"])) and bool(torch.isfinite(enc_g.grad).all()) and float(enc_g.grad.abs().sum()) > 0


def test_voca_train_step_bench_size_features_equal_windows_and_graph_equals_eager():
    """BASELINE configs[1] size (32 sequences x 600 frames): the whole train step is checked through properties that do not
    need the CPU oracle at this size - on-chip windows (features input) vs materialised windows, and hipGraph replay vs eager
    launches, must leave bit-identical losses and parameters after three steps."""
    from a2f_amd.data import feature_windows
    from a2f_amd.trainer import TrainStep
    S, T = 32, 600
    g = torch.Generator().manual_seed(21)
    feats = torch.randn(S, T, 29, generator=g)
    B = S * T
    subj = torch.randint(0, 8, (S,), generator=g)
    oh = torch.zeros(S, 12); oh[torch.arange(S), subj] = 1.0
    oh = oh.repeat_interleave(T, 0).to(DEV)
    templ = (torch.randn(8, 5023, 3, generator=g) * 5)[subj].repeat_interleave(T, 0).to(DEV)
    gg = torch.Generator(device=DEV).manual_seed(22)
    gt = templ + torch.randn(B, 5023, 3, generator=gg, device=DEV) * 0.5
    results = []
    for x, use_graph in ((feature_windows(feats), False), (feats, False), (feature_windows(feats), True)):
        torch.manual_seed(5)
        m = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
        step = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-4, use_graph=use_graph)
        xd = x.to(DEV)
        for _ in range(4 if use_graph else 3):          # the graph path spends its first two calls on eager warm-up + one capture step: all are real steps
            out = step(xd, oh, gt, templ)
        torch.cuda.synchronize()
        results.append((float(out["loss"]), [p.detach().clone() for p in m.parameters()], step))
    (l_w, p_w, _), (l_f, p_f, _), (l_g, p_g, sg) = results
    assert l_w == l_f and all(torch.equal(a, b) for a, b in zip(p_w, p_f))
    # graph path: same arithmetic, one more step taken -> compare after aligning the step count
    torch.manual_seed(5)
    m2 = a2f_amd.Voca(n_verts=15069, n_onehot=12).to(DEV)
    s2 = TrainStep(m2, a2f_amd.VocaLoss(), lr=1e-4, use_graph=False)
    xw = feature_windows(feats).to(DEV)
    for _ in range(4):
        o2 = s2(xw, oh, gt, templ)
    torch.cuda.synchronize()
    assert float(o2["loss"]) == l_g and all(torch.equal(a, b) for a, b in zip(m2.parameters(), p_g))


# ------------------------------------------------------------------------------------------------------------------------
# Full-size configurations with the REAL encoder (BASELINE configs[2] and configs[4]) against the CPU oracle
# ------------------------------------------------------------------------------------------------------------------------
# A second encoder weight set for the every-gradient check: the plain formula weights let the token representations collapse onto a common direction in the
# deep layers (every residual branch adds the same large vector to every frame), so dP - delta in the attention backward cancels and the q / k gradients of layers
# 8-11 are ill-conditioned - the reference's own fp32 arithmetic misses float64 by 2e-4 .. 2e-2 there.  With the two residual-branch output projections scaled
# to a quarter the frames stay diverse and the fp32 reference resolves EVERY gradient to <= 2.5e-5 of float64 at T = 600 (tools/grad_conditioning.py; larger
# q / k gains instead make the whole network chaotic: gain 2 puts every gradient at 1e-3, gain 4 at O(1)) - there the 1e-4 bar applies with no floor at all.
WELL_CONDITIONED_GAINS = {"original0": 40.0, "attention.out_proj": 0.25, "feed_forward.output_dense": 0.25}


def _faceformer_full(gains=None):
    m = _faceformer(stub=False)
    esd = W.formula_state_dict(W.wav2vec2_shapes(), gains=gains or {"original0": 40.0})
    m.audio_encoder.load_state_dict(esd)
    dsd = W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS)
    return m.to(DEV).eval(), esd, dsd


def test_faceformer_config2_full_size_vs_oracle(monkeypatch):
    """BASELINE configs[2] at full size: N = 160 008 samples -> T = 600 frames through the real wav2vec2-base encoder, the
    incremental decoder, the 15069-wide head and FaceFormerLoss; forward <= 1e-4 on vertex offsets, the loss, d loss / d audio and EVERY parameter gradient
    against the oracle's autograd (oracle/wav2vec2.py + decode_incremental; reference: faceformer.py:139-191, wav2vec.py:91-189) - the committed fixture
    g15_c2_formula_s600_subj5 (fp32 oracle: forward, loss; float64 oracle: gradients + the fp32 oracle's own distance from them).
    Runs the path the 64-sequence bench step runs: Transformer products on split-32 image operands (forced: one sequence is 600 rows)."""
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0)
    m, _esd, _dsd = _faceformer_full()
    T = 600
    audio, oh, template, gt = _t600_inputs(600, 5)
    fix = load("g15_c2_formula_s600_subj5")
    a_gpu = audio.to(DEV).requires_grad_(True)
    pred, loss = m.forward_with_loss(a_gpu, oh.to(DEV), template.to(DEV), gt.to(DEV))
    assert pred.shape == (1, T, 5023, 3)
    assert packed_error(fix, "pred_off", pred.detach().cpu() - template.view(1, 1, 5023, 3)) < TOL
    assert abs(float(loss["loss"].detach()) - float(fix["loss32"])) <= 1e-4 * abs(float(fix["loss32"]))
    loss["loss"].backward()
    assert packed_error(fix, "g64::audio", a_gpu.grad) < TOL
    floored = _assert_all_grads_g15(m, fix)
    # the ill-conditioned set is pinned: a gradient that needs the float64 floor and is not on the committed list fails (the list cannot silently grow)
    allowed = {l.strip() for l in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ill_conditioned_gradients_T600.txt")) if l.strip() and not l.startswith("#")}
    assert set(floored) <= allowed, sorted(set(floored) - allowed)


@pytest.mark.parametrize("seed,subject", [(600, 2), (600, 5), (601, 2), (601, 5), (602, 2), (602, 5)])
def test_faceformer_config2_full_size_every_gradient_without_a_floor(seed, subject, monkeypatch):
    """The same full-size step (N = 160 008 samples -> T = 600, real encoder, image-operand path forced) on the WELL-CONDITIONED weight set, SIX input draws
    (fixtures g15_c2_wellcond_*): all 223 parameter gradients and d loss / d audio <= 1e-4, max_floor_share = 0 - no gradient is excused by conditioning.

    The one thing that lifts a draw over the bar without any arithmetic being wrong is a ReLU unit of the decoder's feed-forward on its kink: the gradient is
    discontinuous there, and a pre-activation closer to zero than the arithmetic's FORWARD resolution picks a side.  The HIP path's forward sits 1e-5 ... 4e-5 from the
    oracle (bf16x3 products: 2^-16 each; the bar is 1e-4), a sequence has 600 x 128 = 76 800 units with O(1) pre-activations, so a handful of units per draw lie inside
    that resolution and about half of the draws flip one; the reference's own fp32 arithmetic (resolution ~1e-6) does it on one draw in six.  That is a CHECKED fact
    here, not a comment.  The fixtures list every unit whose float64 pre-activation is within 1e-4 of zero; after the HIP forward the decoder's own saved activations
    (h = relu(.) of every frame, the recurrence's save block) say on which side the HIP path put each of them:
      * no unit flipped (three of the six draws at the time of writing: (600, 2), (600, 5), (601, 5)): every one of the 223 gradients and d loss / d audio <= 1e-4;
      * a flip is DEMONSTRATED (the HIP activation of a listed unit is on the other side than float64; (601, 2): a unit 1.6e-6 from zero, (602, 5): 2.7e-5; (602, 2):
        8e-6 - there the fp32 ORACLE itself flips too and sits 6.5e-4, median over 223 gradients, from float64): the gradients are held to the configuration-level
        1e-3 (measured 2e-4 ... 1e-3; each against the nearer of the float64 and the fp32 reference value, the fixture holds both), and every flipped unit must lie
        within the forward bar (|float64 pre-activation| < 1e-4).
    The forward (vertex offsets) is <= 1e-4 on all six draws."""
    import numpy as np
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0)
    m, _esd, _dsd = _faceformer_full(WELL_CONDITIONED_GAINS)
    audio, oh, template, gt = _t600_inputs(seed, subject)
    fix = load(f"g15_c2_wellcond_s{seed}_subj{subject}")
    T = 600
    erefs = [float(fix[k]) for k in fix if k.startswith("eref::") and not k.endswith(_EXACT_ZERO_GRADS)]
    oracle_kink = float(np.median(erefs)) > TOL
    a_gpu = audio.to(DEV).requires_grad_(True)
    enc = m.encode(a_gpu, T)
    y = m.decode_hidden(enc, oh.to(DEV))
    save = y.grad_fn.saved_tensors[1]                                   # functional.FFDecoderFn: (packed parameters, the recurrence's save block)
    h = save[: T * 896].view(T, 896)[:, 320:448]                        # csrc/ff_decoder.hip: SV_STRIDE = 896 floats per frame, SV_H = 320: relu(linear1) (128)
    fr, un, pre = fix["near_kink_frame"], fix["near_kink_unit"], fix["near_kink_pre64"]
    flips = [(int(f), int(u), float(v)) for f, u, v in zip(fr, un, pre) if (float(h[int(f), int(u)]) > 0.0) != (float(v) > 0.0)]
    print(f"draw (seed {seed}, subject {subject}): float64 margin of the nearest unit {float(fix['relu_margin64']):.3e}; {len(fr)} units within 1e-4 of zero; fp32 oracle's own median "
          f"distance from float64 {float(np.median(erefs)):.2e}; units the HIP path puts on the other side than float64 (frame, unit, float64 pre-activation): {flips}")
    assert oracle_kink == ((seed, subject) == (602, 2))                 # the reference's own fp32 arithmetic flips a unit on exactly this draw
    tol = 1e-3 if flips else TOL                                        # a DEMONSTRATED flip, and nothing else, admits the configuration-level bar
    assert all(abs(v) < 1e-4 for _f, _u, v in flips)                    # ... and only units inside the forward bar may flip
    pred, out3 = m._head_loss_batch(y, template.to(DEV), gt.to(DEV), 1.0, 10.0)
    assert packed_error(fix, "pred_off", pred.detach().cpu().view(1, T, 5023, 3) - template.view(1, 1, 5023, 3)) < TOL
    out3[0].backward()
    print("d loss / d audio vs the oracle:", _g15_error(fix, "audio", a_gpu.grad), "fp32 oracle vs float64:", float(fix["eref::audio"]), "bar:", tol)
    assert _assert_all_grads_g15(m, fix, tol=tol, max_floor_share=0.0) == []
    assert _g15_error(fix, "audio", a_gpu.grad) < tol


# gradients that are identically zero in exact arithmetic (only rounding noise on both sides): k_proj.bias shifts every score of a
# query by the same amount (softmax shift invariance)
_EXACT_ZERO_GRADS = ("attention.k_proj.bias",)
# The split-bf16 products carry 2^-16 per operand where fp32 carries 2^-24: on a gradient that is the small difference of large terms
# (d loss / d q, k of a layer whose attention is nearly uniform: dS = P (dP - delta) with |dP - delta| << |dP|) the HIP path can be up to
# 2^(24 - 16) = 256 times noisier than the reference's fp32 arithmetic - which itself misses the float64 value by up to 2e-2 there (layer
# 11 of the formula weights at T = 600; measured ratios 30 - 80 in round 3, 7 - 27 since the round-4 attention kernels).  Such a gradient is judged against float64 with a floor proportional
# to the REFERENCE's own fp32 error; everywhere the reference resolves a gradient to 4e-7 or better the plain 1e-4 bar applies.
_ILL_CONDITIONED_FLOOR = 64.0          # T = 600: measured ratios 7 - 27 (round 5); the T = 3000 test passes its own (ratios up to 215: five times the keys per softmax row)


def _t600_inputs(seed, subject):
    """= tests/golden/gen_golden_fullsize.py::inputs_T600 (the fixtures G15 store no inputs: both sides regenerate them from the seed)."""
    T = 600
    gen = torch.Generator().manual_seed(seed)
    audio = torch.randn(1, T * 16000 // 60 + 8, generator=gen) * 0.1
    oh = torch.zeros(1, 12); oh[0, subject] = 1
    template = torch.randn(1, 5023, 3, generator=gen) * 5
    gt = template.view(1, 1, 5023, 3) + torch.randn(1, T, 5023, 3, generator=gen) * 0.5
    return audio, oh, template, gt


def _g15_error(fix, key, t):
    """Distance of a HIP gradient from the oracle: from the float64 value, or - where the fixture also holds the fp32 oracle's value (it does wherever the reference's
    own fp32 arithmetic sits more than 2e-5 from float64) - from whichever of the two is nearer: the parity bar is defined on the fp32 reference, float64 is the truth
    both are measured by."""
    e = packed_error(fix, "g64::" + key, t)
    if ("g32::" + key) in fix or ("g32::" + key + "::l2") in fix:
        e = min(e, packed_error(fix, "g32::" + key, t))
    return e


def _assert_all_grads_g15(m, fix, tol=TOL, max_floor_share=0.15, floor=_ILL_CONDITIONED_FLOOR):
    """EVERY parameter gradient of the HIP path against a full-size oracle fixture (G15, tests/golden/gen_golden_fullsize.py): <= tol from the oracle's gradient
    (_g15_error), or - ill-conditioned gradients, where the fp32 ORACLE itself sits `eref` from float64 - <= `floor` x eref from float64 (_ILL_CONDITIONED_FLOOR above).
    Wherever the fp32 oracle resolves a gradient (eref ~ 1e-6) the float64 value and the fp32 one are the same number at the bar's resolution."""
    params = dict(m.named_parameters())
    rows, floored = [], []
    for key in sorted(k[len("eref::"):] for k in fix if k.startswith("eref::")):
        if key in ("audio", "enc") or key not in params or key.endswith(_EXACT_ZERO_GRADS):
            continue
        l2 = float(fix["g64::" + key + "::l2"]) if ("g64::" + key + "::l2") in fix else float(abs(fix["g64::" + key]).max())
        if l2 == 0.0:
            continue
        e, eref = _g15_error(fix, key, params[key].grad), float(fix["eref::" + key])
        if e < tol:
            rows.append((e, key))
            continue
        e64 = packed_error(fix, "g64::" + key, params[key].grad)
        floored.append((key, f"hip {e64:.2e}", f"fp32 oracle {eref:.2e}", f"ratio {e64 / eref:.0f}"))
        assert e64 <= max(tol, floor * eref), (key, e64, eref)
    rows.sort(reverse=True)
    print("gradients compared:", len(rows) + len(floored), "worst well-conditioned:", [(f"{e:.2e}", k) for e, k in rows[:8]])
    print("ill-conditioned (judged with the float64 floor):", floored)
    assert len(rows) + len(floored) >= 10 and len(floored) <= max_floor_share * (len(rows) + len(floored))
    return [f[0] for f in floored]


def _ff_long_inputs(T, seed):
    gen = torch.Generator().manual_seed(seed)
    audio = torch.randn(1, T * 16000 // 60 + 8, generator=gen) * 0.1
    oh = torch.zeros(1, 12); oh[0, 2] = 1
    template = torch.randn(1, 5023, 3, generator=gen) * 5
    gt = template.view(1, 1, 5023, 3) + torch.randn(1, T, 5023, 3, generator=gen) * 0.5
    return audio, oh, template, gt


def test_faceformer_config5_T3000_decoder_gradients_vs_oracle():
    """BASELINE configs[4]'s length through the blocked backward: T = 3000 = 24 blocks of 128 steps (cross-block dK / dV kernels, prep
    pass, per-block recurrence launches) with a pinned encoder output - EVERY decoder-side gradient and d loss / d encoder-output
    against the oracle's autograd through the whole 3000-step recurrence (faceformer.py:154-185): fixture g15_c5_T3000_decoder_s3001."""
    T = 3000
    m = _faceformer()
    audio, oh, template, gt = _ff_long_inputs(T, 3001)
    enc = W.formula_input("t3000.enc", (1, T, 768))
    fix = load("g15_c5_T3000_decoder_s3001")
    e_gpu = enc.to(DEV).requires_grad_(True)
    pred, loss = m.forward_with_loss(audio.to(DEV), oh.to(DEV), template.to(DEV), gt.to(DEV), encoder_states=e_gpu)
    assert packed_error(fix, "pred_off", pred.detach().cpu() - template.view(1, 1, 5023, 3)) < TOL
    assert abs(float(loss["loss"].detach()) - float(fix["loss32"])) <= 1e-4 * abs(float(fix["loss32"]))
    loss["loss"].backward()
    assert packed_error(fix, "g64::enc", e_gpu.grad) < TOL
    _assert_all_grads_g15(m, fix, max_floor_share=0.0)


def test_faceformer_config5_T3000_full_gradients_vs_oracle():
    """configs[4] end to end with the real encoder: N = 800 008 samples -> 3000 frames (5x beyond the reference's 600-frame mask / PPE tables,
    faceformer.py:118-120), forward + backward through the decoder's 24 blocks, the flash-attention backward over 3000 keys and the whole encoder, in the
    configuration's parity setting (bf16x3): the encoder output, the vertex offsets, d loss / d audio and every parameter gradient against the oracle's autograd
    (fixture g15_c5_T3000_full_s3002) at the fp32-class 1e-4 - ten times inside the configuration's stated 1e-3."""
    m, _esd, _dsd = _faceformer_full()
    T = 3000
    audio, oh, template, gt = _ff_long_inputs(T, 3002)
    fix = load("g15_c5_T3000_full_s3002")
    with torch.no_grad():
        assert packed_error(fix, "enc", m.encode(audio.to(DEV), T)) < TOL
    a_gpu = audio.to(DEV).requires_grad_(True)
    pred, loss = m.forward_with_loss(a_gpu, oh.to(DEV), template.to(DEV), gt.to(DEV))
    assert pred.shape == (1, T, 5023, 3)
    err = packed_error(fix, "pred_off", pred.detach().cpu() - template.view(1, 1, 5023, 3))
    print("T=3000 parity setting, vertex offsets vs oracle:", err)
    assert err < TOL
    loss["loss"].backward()
    assert _g15_error(fix, "audio", a_gpu.grad) < TOL                   # (at 3000 frames the fp32 oracle's own d loss / d audio is 1e-4-class from float64: the nearer of the two)
    _assert_all_grads_g15(m, fix, floor=256.0)                          # q / k gradients under near-uniform attention over 3000 keys: measured ratios up to 215


def test_faceformer_config5_T3000_other_precision_settings_miss_the_tolerance():
    """bench.py's configs[4] entry is timed in the library's DEFAULT setting (bf16x3 GEMMs and attention) - the test above, at 10x inside the configuration's
    1e-3.  The cheaper settings, measured against the same fixture: single-term e4m3 attention misses the 1e-3 on the vertex offsets (bench.py carries it as a
    labelled variant with its error); plain-bf16 attention under bf16x3 GEMMs meets it FORWARD (1-3e-4) but not on the gradients (d loss / d audio 2.5e-3, 49-156
    of 235 parameter gradients above 1e-3: tools/attn_precision_probe.py) - so neither is a configs[4] setting, and if one ever were this test and bench.py's
    labelling must change."""
    from a2f_amd import ops
    m, _esd, _dsd = _faceformer_full()
    T = 3000
    audio, oh, template, gt = _ff_long_inputs(T, 3002)
    fix = load("g15_c5_T3000_full_s3002")
    be = ops.backend()
    assert be.mm_precision == 1 and be.attn_precision is None          # what the timed bench entry runs
    errs, gerr = {}, {}
    for name, (mmp, ap) in {"bf16_everywhere": (0, None), "bf16x3_gemms_fp8_attention": (1, 2), "bf16x3_gemms_bf16_attention": (1, 0)}.items():
        old = be.mm_precision, be.attn_precision
        be.mm_precision, be.attn_precision = mmp, ap
        try:
            m.zero_grad(set_to_none=True)
            a_gpu = audio.to(DEV).requires_grad_(True)
            p2, l2 = m.forward_with_loss(a_gpu, oh.to(DEV), template.to(DEV), gt.to(DEV))
            l2["loss"].backward()
        finally:
            be.mm_precision, be.attn_precision = old
        errs[name] = packed_error(fix, "pred_off", p2.detach().cpu() - template.view(1, 1, 5023, 3))
        gerr[name] = packed_error(fix, "g64::audio", a_gpu.grad)
    print("T=3000 vertex-offset error vs oracle:", errs, "d loss / d audio:", gerr)
    assert errs["bf16x3_gemms_fp8_attention"] > 1e-3
    assert errs["bf16x3_gemms_bf16_attention"] < 1e-3 < gerr["bf16x3_gemms_bf16_attention"]


@pytest.mark.parametrize("force_pipeline", [True])
def test_bench_self_launches_two_ranks(force_pipeline):
    """`python bench.py --gpus 2` with no torchrun environment (as the driver starts it): the script must launch the two
    ranks itself as a child torchrun before touching the GPU, relay ONE JSON line and exit 0.  Rehearsal mode on the one-GPU
    box: both ranks share cuda:0, collectives over gloo.  (Only the pipelined FaceFormer form runs here, one timed step after the warm-up: over gloo the 385-MB gradient exchange of one step takes ~13 s; the single-pass
    bucketed exchange is what tests/test_distributed_cpu.py runs at world 2 / 4 / 8.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(A2F_BENCH_BACKEND="gloo", A2F_BENCH_SHARE_GPU="1")
    if force_pipeline:
        # the two-half-batch FaceFormer step (side streams) at small shapes: the bucketed gradient exchange is then armed before the LAST encoder
        # backward only (trainer.BucketedAllReduce.begin via before_last_backward) and its collectives are issued from autograd hooks on the main stream
        env.update(A2F_FF_PIPELINE_MIN_FRAMES="60", A2F_FF_PIPELINE_MIN_SEQS="0")
    # the secondary entries run too (small shapes): every rank must issue the same collectives in every leg, the rank-0-only roofline passes included
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--seqs", "4",
                        "--frames", "64", "--ff-seqs", "2", "--ff-frames", "60", "--ff-long-seqs", "2", "--ff-long-frames", "120", "--a2m-batch", "64",
                        "--secondary-steps", "1", "--no-batch-points", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=root)      # (the CPU-baseline leg is rank 0 alone and a minute of host time: the default bench run covers it)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    assert {e["model"] for e in out["secondary"]} == {"faceformer", "audio2mesh", "faceformer_long"}
    assert all(e["config"]["parallelism"] == "dp2" and e["value"] > 0 for e in out["secondary"])
    assert out["distributed"]["rccl_world"] == 2 and "expected 3" in out["distributed"]["allreduce_selfcheck"] and len(out["distributed"]["ranks"]) == 2
    ff = next(e for e in out["secondary"] if e["model"] == "faceformer")
    assert "buckets" in ff["config"]["gradient_exchange"]


def test_bench_secondary_lines_small():
    """The driver-run bench line carries FaceFormer (configs[2]), Audio2Mesh and the long-context FaceFormer (configs[4]) as `secondary`
    entries, each with its own roofline (small shapes here; the default run uses 64 x 600 frames, B = 1024 and 12 x 3000 frames)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "3", "--seqs", "4", "--frames", "64",
                        "--ff-seqs", "2", "--ff-frames", "60", "--ff-long-seqs", "1", "--ff-long-frames", "120", "--a2m-batch", "64",
                        "--secondary-steps", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["roofline"]["kernel"] == "head_x3_kernel" and out["roofline"]["measured_stream_GBs"]["copy"] > 1000
    assert out["timing"]["blocks"] >= 1 and len(out["timing"]["ms_per_step_per_block"]) == out["timing"]["blocks"]
    sec = {e["model"]: e for e in out["secondary"]}
    assert set(sec) == {"faceformer", "audio2mesh", "faceformer_long"}                   # configs[2], Audio2Mesh, configs[4]
    # configs[4] is timed in the setting that meets its tolerance (bf16x3); fp8 attention only appears as labelled variants with their error
    assert "fp8" not in sec["faceformer_long"]["dtype"] and "fp8" not in sec["faceformer"]["dtype"]
    assert sec["faceformer_long"]["parity_setting"]["tolerance"] == 1e-3
    for v in sec["faceformer_long"]["variants"]:
        assert "fp8" in v["variant"] and v["vertex_offset_rel_l2_vs_parity_setting"] > 0 and v["value"] > 0
        assert v["meets_1e-3"] == (v["vertex_offset_rel_l2_vs_parity_setting"] <= 1e-3)
    assert {p["seqs_per_step"] for p in sec["faceformer"]["sequences_per_step_points"]} == {1}       # S = 2 here: only the S = 1 point is below it
    for e in sec.values():
        r = e["roofline"]
        assert e["value"] > 0 and r["bound"] == "mfma" and r["peak"] == 2500.0 and 0 < r["frac"] < 1
        assert abs(r["achieved"] - r["executed_per_algorithmic"] * r["algorithmic"]) <= 0.02 * r["achieved"] + 0.02
    k = sec["faceformer"]["roofline"]["kernel"]
    assert k["name"].startswith("gemm_mfma_kernel") and k["launches_timed"] > 0 and 0 < k["frac"] < 1


def test_bench_roofline_fractions_stay_below_one_at_the_default_shapes():
    """The formula of the FaceFormer / Audio2Mesh roofline entries at the DEFAULT shapes' measured rates (round 2's driver line: 247.5 k
    frames/s at 64 x 600, 126 k at B = 1024): executed flops against the bf16 peak, never above 1 (round 2 printed 1.35 against the fp32 peak)."""
    for fps, mflop, mult in ((247548.8, 857.0, 3.0), (126000.0, 393.0, 6.0), (400000.0, 857.0, 3.0)):
        alg = fps * mflop * 1e6 / 1e12
        assert 0 < mult * alg / 2500.0 < 1


def test_mixed_precision_backward_runs_in_the_forward_precision():
    """`percision: 16-mixed` (encoder_autocast): the encoder's backward nodes run in the precision their forward ran in (plain bf16), although
    Faceformer.encode restores the backend's setting right after the forward: gradients equal, bit for bit, those of a step whose whole
    forward + backward ran under mm_precision = 0 - and differ from the bf16-forward / bf16x3-backward mixture round 2 computed."""
    from a2f_amd import ops
    be = ops.backend()
    torch.manual_seed(0)
    m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12).to(DEV).eval()
    with torch.no_grad():
        m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
    T = 40
    audio = torch.randn(2, T * 16000 // 60 + 8, device=DEV) * 0.1

    def grads(autocast, held):
        m.zero_grad()
        m.encoder_autocast = autocast
        old = be.mm_precision
        if held is not None:
            be.mm_precision = held
        try:
            h = m.encode(audio, T)
            if held is None:
                assert be.mm_precision == old                      # encode restored the setting before the backward runs
            (h * torch.linspace(-1, 1, h.numel(), device=DEV).view_as(h)).sum().backward()
        finally:
            be.mm_precision = old
            m.encoder_autocast = None
        return [p.grad.clone() for p in m.audio_encoder.parameters() if p.grad is not None]

    g_auto = grads(torch.bfloat16, None)
    g_held = grads(None, 0)
    g_split = grads(None, None)
    assert all(torch.equal(a, b) for a, b in zip(g_auto, g_held))
    assert any(not torch.equal(a, b) for a, b in zip(g_auto, g_split))


# ------------------------------------------------------------------------------------------------------------------------
# N1: the Lightning-free trainer against the reference-generated training trajectory (G10) and the fit loop on the GPU
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_graph", [False, True])
def test_trainstep_training_trajectory_g10(use_graph):
    """Three TrainStep steps on the HIP path (x100 unpack, fused forward / loss / backward, flat Adam, weight decay lr/10)
    reproduce the loss / err trajectory and the final parameters of the reference VOCA model + VocaLoss + torch.optim.Adam
    (fixture G10, generated by tests/golden/gen_golden.py from /root/reference)."""
    from a2f_amd.trainer import TrainStep
    g = load("g10_voca_train_trajectory")
    m, _ = _voca()
    step = TrainStep(m, a2f_amd.VocaLoss(), lr=float(g["lr"]), use_graph=use_graph)
    B = int(g["B"])
    static = None
    for it in range(3):
        tv = W.formula_input(f"g10.template{it}", (B, 5023, 3), 0.05)
        batch = {"audio": W.formula_input(f"g10.x{it}", (B, 29, 16)).to(DEV), "one_hot": torch.from_numpy(g["one_hot"]).to(DEV),
                 "template_vert": tv.to(DEV), "verts": (tv + W.formula_input(f"g10.gt{it}", (B, 5023, 3), 0.005)).to(DEV)}
        ins = TrainStep.unpack_batch(batch)
        if use_graph:                                   # graph replay: same input buffers every step
            if static is None:
                static = [t.clone() for t in ins]
            for s_, t_ in zip(static, ins):
                s_.copy_(t_)
            ins = static
        out = step(*ins)
        for k in ("loss", "rec_loss", "vel_loss", "err"):
            ref = float(g[f"step{it}::{k}"])
            assert abs(float(out[k]) - ref) <= 5e-5 * abs(ref), (it, k, float(out[k]), ref)
    for k, p in m.named_parameters():
        assert packed_error(g, "final::" + k, p) < 1e-4, k


def test_fit_loop_on_gpu_voca_and_faceformer(tmp_path):
    """a2f_amd.fit on the GPU: VOCA two epochs (validation, best-only checkpoint, reload) and the FaceFormer branch of
    train.py:30-34 (batch 1, whole clips) for one epoch on a 30-frame clip."""
    from a2f_amd import fit as F
    from a2f_amd.registry import ExpConfig
    from a2f_amd.train import synthetic_batch
    base = dict(batch_size=8, modelname="voca", vertex_count=NV, one_hot_size=12, split_frame=True, percision="32", lr=1e-3,
                feature_extractor=None, sample_rate=22000, n_feature=16, out_dim=29, win_length=440)
    for name, epochs in (("voca", 2), ("faceformer", 1)):
        cfg = F.apply_model_overrides(ExpConfig(**dict(base, modelname=name)))
        torch.manual_seed(0)
        tr = lambda e: (synthetic_batch(cfg, cfg.batch_size, DEV, 100 + 3 * e + i, frames=30) for i in range(3))
        va = lambda e: (synthetic_batch(cfg, cfg.batch_size, DEV, 7, frames=30),)
        model, hist = F.fit(cfg, tr, va, DEV, str(tmp_path / name), max_epochs=epochs, log=lambda *_: None)
        assert len(hist) == epochs and all(h["val/err"] == h["val/err"] and h["val/err"] > 0 for h in hist)
        files = os.listdir(tmp_path / name / "version_0" / "checkpoints")
        assert len(files) == 1 and files[0].startswith("epoch=")
        ck = torch.load(tmp_path / name / "version_0" / "checkpoints" / files[0], weights_only=False)
        assert ck["hyper_parameters"]["config"].modelname == name and all(k.startswith("model.") for k in ck["state_dict"])


def test_wav2vec_extractor_on_gpu_g12():
    """N2: Wav2VecExtractor with its encoder on the hand-written HIP path (plain 49 Hz forward, no interpolation) against the
    HF-generated fixture G12."""
    from a2f_amd.extractor import Wav2VecExtractor
    g = load("g12_wav2vec_extractor")
    for name in ("a", "b"):
        B, n, out_dim, n_feature = (int(v) for v in g[f"{name}::dims"])
        ex = Wav2VecExtractor(sample_rate=16000, n_feature=n_feature, out_dim=out_dim).eval()
        ex.model.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0}))
        ex.to(DEV)
        x = (W.formula_input(f"g12.audio.{name}", (B, n), 0.1) + 0.02 * torch.arange(B).view(B, 1)).to(DEV)
        with torch.no_grad():
            y = ex(x)
        assert y.is_cuda and tuple(y.shape) == (B, out_dim, n_feature)
        assert packed_error(g, f"{name}::features", y) < TOL


def test_trainstep_falls_back_to_eager_when_graph_capture_fails(monkeypatch):
    """TrainStep(use_graph=True): a failed hipGraph capture must not abort training - the step continues with eager launches
    of the same kernels and produces the same numbers as a step that never tried to capture."""
    from a2f_amd.trainer import TrainStep
    gen = torch.Generator().manual_seed(11)
    B = 6
    x = torch.randn(B, 29, 16, generator=gen).to(DEV)
    oh = torch.zeros(B, 12); oh[torch.arange(B), torch.arange(B) % 8] = 1
    t = (torch.randn(B, 5023, 3, generator=gen) * 5).to(DEV)
    gt = t + (torch.randn(B, 5023, 3, generator=gen) * 0.5).to(DEV)
    losses = {}
    for mode in ("eager", "broken_graph"):
        m, _ = _voca()
        step = TrainStep(m, a2f_amd.VocaLoss(), lr=1e-3, use_graph=(mode != "eager"))
        if mode == "broken_graph":
            class Boom:
                def __init__(self, *a, **k):
                    raise RuntimeError("capture refused (test)")
            monkeypatch.setattr(torch.cuda, "graph", Boom)
        with (pytest.warns(UserWarning, match="capture") if mode == "broken_graph" else __import__("contextlib").nullcontext()):
            losses[mode] = [float(step(x, oh.to(DEV), gt, t)["loss"]) for _ in range(4)]      # steps 1-2 warm up, step 3 tries to capture
        if mode == "broken_graph":
            assert step.use_graph is False
            monkeypatch.undo()
    assert losses["eager"] == losses["broken_graph"]


@pytest.mark.parametrize("T", [40, 41])
def test_faceformer_batch_step_one_head_launch_equals_per_sequence_steps(T):
    """forward_with_loss_batch: S sequences with S different subject templates through ONE fused-head launch (even T; template row
    = frame // T inside the kernel) against the mean of S batch-1 `forward_with_loss` calls, the reference's own formulation
    (train.py:33 batch = 1; loss.py:4-17).  Odd T takes the per-sequence launches (each drops its own tail frame)."""
    import types
    S = 3

    class _Enc(torch.nn.Module):
        def forward(self, wave, dataset, frame_num=None):
            return types.SimpleNamespace(last_hidden_state=self.states[: wave.shape[0]] if wave.shape[0] == S else self.states[self.pick:self.pick + 1])

    m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12, audio_encoder=_Enc())
    m.load_state_dict(W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS), strict=False)
    m.to(DEV).eval()
    enc = W.formula_input("batchhead.enc", (S, T, 768)).to(DEV).requires_grad_(True)
    m.audio_encoder.states = enc
    audio = torch.zeros(S, T * 16000 // 60 + 8, device=DEV)
    oh = torch.zeros(S, 12, device=DEV); oh[0, 1] = oh[1, 7] = oh[2, 4] = 1
    template = W.formula_input("batchhead.template", (S, 5023, 3), 5.0).to(DEV)
    gt = template.view(S, 1, 5023, 3) + W.formula_input("batchhead.gt", (S, T, 5023, 3), 0.5).to(DEV)
    pred_b, loss_b = m.forward_with_loss_batch(audio, oh, template, gt)
    loss_b["loss"].backward()
    gb = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    ge_b = enc.grad.clone()
    m.zero_grad(); enc.grad = None
    tot, preds = 0.0, []
    for s in range(S):
        m.audio_encoder.pick = s
        p, l = m.forward_with_loss(audio[s:s + 1], oh[s:s + 1], template[s:s + 1], gt[s:s + 1])
        (l["loss"] / S).backward()
        tot = tot + l["loss"].detach() / S
        preds.append(p)
    assert rel_l2(pred_b.cpu(), torch.cat(preds).cpu()) < 1e-6
    assert abs(float(loss_b["loss"]) - float(tot)) <= 2e-5 * abs(float(tot))
    assert rel_l2(ge_b.cpu(), enc.grad.cpu()) < TOL
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert rel_l2(gb[k].cpu(), p.grad.cpu()) < TOL, k


@pytest.mark.parametrize("S,T,parts", [(3, 30, 2), (4, 61, 2), (5, 30, 4), (9, 24, 4)])
def test_faceformer_pipelined_step_equals_single_stream_step(S, T, parts, monkeypatch):
    """Faceformer.train_step_pipelined (two or four parts of the batch on two HIP streams, decoder recurrence under the other parts' encoder work):
    BITWISE equal to the same parts run one after the other on one stream (same kernels, same accumulation order - any
    missing stream dependency or allocator reuse hazard shows as a difference), twice in a row (the second step reuses freed
    blocks); and equal within the summation-order noise of the weight-gradient GEMMs to the unsplit batch step."""
    m, _, _ = _faceformer_full()
    m.pipeline_parts = parts                                                    # two parts is the shipped setting; the step is written for any number
    m.train()
    m.audio_encoder.train_regularisers = m.decoder_regularisers = False          # every call of this comparison must compute the same function
    gen = torch.Generator().manual_seed(77)
    audio = (torch.randn(S, T * 16000 // 60 + 8, generator=gen) * 0.1).to(DEV)
    oh = torch.zeros(S, 12); oh[torch.arange(S), torch.arange(S) % 8] = 1
    template = torch.randn(S, 5023, 3, generator=gen) * 5
    gt = (template.view(S, 1, 5023, 3) + torch.randn(S, T, 5023, 3, generator=gen) * 0.5).to(DEV)
    oh, template = oh.to(DEV), template.to(DEV)
    m.zero_grad()
    _p, loss = m.forward_with_loss_batch(audio, oh, template, gt)
    loss["loss"].backward()
    whole = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    cuts = [(S * g + parts - 1) // parts for g in range(parts)] + [S]
    seq_loss = 0.0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        _p, l = m.forward_with_loss_batch(audio[lo:hi], oh[lo:hi], template[lo:hi], gt[lo:hi])
        (l["loss"] * ((hi - lo) / S)).backward()
        seq_loss = seq_loss + l["loss"].detach() * ((hi - lo) / S)
    halves = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    # The unsplit step is the same function to rounding, but not to 1e-4 in every case: encoder states that differ at the 1e-5 level (other tiles, other
    # summation order at another batch size) can put ONE ReLU unit of the decoder's feed-forward layer on the other side of zero, which moves that
    # sequence's gradients by ~2e-2 (tools/ff_short_diag2.py: sequence 3 of the (9, 24) case - alone 2.1e-2 from the float64 oracle, inside the batch
    # 2.5e-5; every other sequence <= 1e-4 both ways).  The four-part cases hit such a unit, so they assert the bitwise half only.
    whole_comparable = parts == 2
    for _rep in range(2):
        m.zero_grad()
        lp = m.train_step_pipelined(audio, oh, template, gt)
        torch.cuda.synchronize()
        assert float(lp["loss"]) == float(seq_loss)
        assert abs(float(lp["loss"]) - float(loss["loss"].detach())) <= 1e-5 * abs(float(loss["loss"].detach()))
        got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        assert set(got) == set(halves)
        for k in halves:
            assert torch.equal(got[k], halves[k]), k
            # against the unsplit step: q / k projection gradients of the deeper layers are ~1e-5 in magnitude with this synthetic
            # model (near-uniform attention; k_proj.bias is analytically zero) and carry only summation-order noise - held as part of
            # the whole gradient vector below, every other tensor individually
            if ".q_proj." not in k and ".k_proj." not in k and whole_comparable:
                assert rel_l2(got[k].cpu(), whole[k].cpu()) < TOL, k
        keys = sorted(halves)
        if whole_comparable:
            # two evaluations of one function whose bf16x3 products are cut differently along K at the two batch sizes (few-row products are split
            # along K automatically, ops.SMALL_M_SPLIT): 2.3e-5 measured for (3, 30, 2), each side ~1e-5 from the float64 oracle
            assert rel_l2(torch.cat([got[k].flatten() for k in keys]).cpu(), torch.cat([whole[k].flatten() for k in keys]).cpu()) < 5e-5


def test_weight_image_cache_follows_the_weights(monkeypatch):
    """encoder_ops._layer_weight_images: a layer's fused q|k|v weight and split-32 weight images are built once per state of the weights.  Three training
    steps (the raw-pointer Adam update between them bumps WEIGHTS_EPOCH), then an in-place torch edit of one weight (tensor version): every loss, the
    parameters after the steps and the prediction after the edit are BITWISE those of the same run with the cache switched off."""
    from a2f_amd import encoder_ops
    from a2f_amd.trainer import TrainStep
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0)
    S, T = 2, 30
    gen = torch.Generator().manual_seed(9)
    audio = (torch.randn(S, T * 16000 // 60 + 8, generator=gen) * 0.1).to(DEV)
    oh = torch.zeros(S, 12); oh[torch.arange(S), torch.arange(S) % 8] = 1
    template = torch.randn(S, 5023, 3, generator=gen) * 5
    gt = (template.view(S, 1, 5023, 3) + torch.randn(S, T, 5023, 3, generator=gen) * 0.5).to(DEV)
    oh, template = oh.to(DEV), template.to(DEV)
    runs = []
    for cache in (True, False):
        monkeypatch.setattr(encoder_ops, "WIMG_CACHE", cache)
        monkeypatch.setattr(encoder_ops, "WIMG_VERIFY", cache)               # the debug guard on: every hit re-splits and compares (must stay silent here)
        m, _, _ = _faceformer_full()
        m.train()
        m.audio_encoder.train_regularisers = m.decoder_regularisers = False
        step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-3)
        losses = [float(step(audio, oh, gt, template)["loss"]) for _ in range(3)]
        # the entry lives on the layer's q_proj.weight parameter (it dies with the model; no process-wide table)
        assert all(("_a2f_wimg" in lay.attention.q_proj.weight.__dict__) == cache for lay in m.audio_encoder.encoder.layers)
        params = [p.detach().clone() for p in m.parameters()]
        with torch.no_grad():
            m.audio_encoder.encoder.layers[3].feed_forward.intermediate_dense.weight.mul_(1.25)          # a torch in-place edit: version counter
            pred = m.eval()(audio[:1], oh[:1], template[:1]).clone()
        runs.append((losses, params, pred))
    (l1, p1, y1), (l0, p0, y0) = runs
    assert l1 == l0 and l1[0] != l1[1] != l1[2]
    assert all(torch.equal(a, b) for a, b in zip(p1, p0))
    assert torch.equal(y1, y0)
    # what the version counters cannot see - a write through `.data` - is what A2F_WIMG_VERIFY exists for: it raises instead of training on stale images
    monkeypatch.setattr(encoder_ops, "WIMG_CACHE", True)
    monkeypatch.setattr(encoder_ops, "WIMG_VERIFY", False)
    m, _, _ = _faceformer_full()
    m.eval()
    with torch.no_grad():
        m(audio[:1], oh[:1], template[:1])                                    # fills the entries
        m.audio_encoder.encoder.layers[0].attention.out_proj.weight.data.mul_(1.5)
        monkeypatch.setattr(encoder_ops, "WIMG_VERIFY", True)
        with pytest.raises(a2f_amd.A2FError, match="stale weight images"):
            m(audio[:1], oh[:1], template[:1])
        encoder_ops.bump_weights_epoch()                                      # what a caller does after such a write
        m(audio[:1], oh[:1], template[:1])


@pytest.mark.parametrize("images", [False, True])
def test_wav2vec2_train_mode_with_reference_masks_g13(images, monkeypatch):
    """The hand-written encoder in TRAIN mode on the GPU, fed every mask the reference drew in its own train-mode forward (fixture
    G13: dropout at the 6 kinds of sites, LayerDrop of layers 3 and 8, SpecAugment rows): last_hidden_state and the recorded
    parameter gradients of the reference within the parity bar.  images: masks inside the image GEMM epilogues / masked LayerNorm backward."""
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0 if images else 1 << 30)
    from test_train_regularisers_cpu import check_against_g13, g13_model_and_input
    g, m, wave, S, T = g13_model_and_input(DEV)
    hs = m(wave, "vocaset", frame_num=T).last_hidden_state
    assert m.last_masks is m.injected_masks
    check_against_g13(g, m, hs, T, TOL, TOL)


def test_wav2vec2_drawn_masks_on_gpu():
    """Masks drawn on the device from torch's generator: reproducible from the seed, different between steps, absent in eval mode;
    about one element in ten is dropped at each site."""
    from a2f_amd.dropout import unpack_keep_bits
    m, _, _ = _faceformer_full()
    enc = m.audio_encoder.train()
    T = 30
    wave = torch.randn(2, T * 16000 // 60 + 8, generator=torch.Generator().manual_seed(3)).to(DEV)
    torch.manual_seed(5)
    a = enc(wave, "vocaset", frame_num=T).last_hidden_state.detach()
    masks = enc.last_masks
    torch.manual_seed(5)
    b = enc(wave, "vocaset", frame_num=T).last_hidden_state.detach()
    c = enc(wave, "vocaset", frame_num=T).last_hidden_state.detach()
    assert torch.equal(a, b) and not torch.equal(a, c)
    for bits, cols in ((masks.featproj[0], 768), (masks.layers[0]["act"][0], 3072), (masks.layers[5]["attn"][0], T)):
        assert abs(float(unpack_keep_bits(bits, cols).float().mean()) - 0.9) < 0.02
    assert masks.spec is not None and int(masks.spec.sum()) >= 2 * 10
    enc.eval()
    d = enc(wave, "vocaset", frame_num=T).last_hidden_state.detach()
    assert enc.last_masks is None and rel_l2(a, d) > 1e-3


@pytest.mark.parametrize("T", [7, 61])
def test_faceformer_decoder_train_mode_per_row_masks_g14(T):
    """ff_decoder_fwd_drop / bwd_drop + the masked cross rows + the head on the GPU in train mode, fed the per-row masks of fixture G14
    (the reference's O(T^3) loop run with its dropout calls replaced by row-consistent masks): predictions, loss terms, every
    decoder-side gradient and d loss / d encoder-output within the parity bar; plain forward + loss module and the fused-loss path."""
    from test_train_regularisers_cpu import check_against_g14, g14_case
    for fused in (False, True):
        g, m, pre, enc, audio, one_hot, template, gt = g14_case(T, DEV)
        if fused:
            pred, loss = m.forward_with_loss(audio, one_hot, template, gt, encoder_states=enc)
        else:
            pred = m(audio, one_hot, template, encoder_states=enc)
            loss = a2f_amd.FaceFormerLoss()(pred, gt)
        assert m.last_decoder_masks is m.injected_decoder_masks
        check_against_g14(g, m, pre, enc, pred, loss, TOL)


def test_faceformer_decoder_drawn_masks_on_gpu():
    """Decoder masks drawn on the device: reproducible from torch's seed, new every step, none in eval mode, keep rate 0.9; the
    decoder kernels with masks stay bitwise deterministic run to run."""
    from a2f_amd.dropout import unpack_keep_bits
    m = _faceformer().train()
    S, T = 3, 40
    enc = W.formula_input("drawn.enc", (S, T, 768)).to(DEV)
    oh = torch.zeros(S, 12, device=DEV); oh[:, 2] = 1
    torch.manual_seed(9)
    a = m.decode_hidden(enc, oh).detach()
    dm = m.last_decoder_masks
    torch.manual_seed(9)
    b = m.decode_hidden(enc, oh).detach()
    c = m.decode_hidden(enc, oh).detach()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(unpack_keep_bits(dm.row, 320).float().mean()) - 0.9) < 0.02
    assert abs(float(unpack_keep_bits(dm.attn, T).float().mean()) - 0.9) < 0.02
    heads = unpack_keep_bits(dm.cross_v, 64).view(S * T, 4, 16)
    assert bool((heads == heads[:, :, :1]).all())                    # one decision per (frame, head)
    m.eval()
    d = m.decode_hidden(enc, oh).detach()
    assert m.last_decoder_masks is None and rel_l2(a.cpu(), d.cpu()) > 1e-3


def test_mfcc_extractor_hip_products_vs_torch_ops():
    """MFCCExtractor on HIP tensors (STFT as a windowed-DFT matrix product, mel filter bank and DCT on a2f_gemm_f32, exact fp32 MFMA)
    against its torch-ops form (torch.stft: rocFFT on the device, and the CPU): same features to fp32 rounding.  PARITY UNPINNED
    against torchaudio (absent here) - this pins the two evaluations of the restated definition against each other."""
    from a2f_amd.extractor import MFCCExtractor
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(8, 11440, generator=gen) * 0.1
    x[3] *= 1e-3
    t = torch.arange(11440) / 22000.0
    x[5] = 0.3 * torch.sin(2 * 3.14159265 * 440.0 * t) + 1e-4 * x[5]          # a tone 70 dB above its noise floor: wide dynamic range
    from torch_ops_mfcc import mfcc_torch_ops
    ex = MFCCExtractor(22000, 32, 52, 440, None, 1024).to(DEV)
    hip = ex(x.to(DEV))
    fft_gpu = mfcc_torch_ops(ex, x.to(DEV))
    cpu = mfcc_torch_ops(MFCCExtractor(22000, 32, 52, 440, None, 1024), x)
    with pytest.raises(a2f_amd.A2FError):
        ex(x)                                                             # no CPU path in the product
    assert hip.shape == (8, 52, 32)
    assert rel_l2(hip.cpu(), cpu) < 1e-4 and rel_l2(hip.cpu(), fft_gpu.cpu()) < 1e-4


@pytest.mark.parametrize("S,T", [(48, 600), (8, 3000)])
def test_faceformer_full_size_pipelined_step_is_bitwise_reproducible(S, T):
    """The train step at bench scale (48 sequences x 600 frames: two half-batches, decoder side on a side stream beside encoder work, weight gradients on
    their lane, masks drawn on the device) run twice from the same state and seed: parameters after two optimizer steps must be BITWISE equal.  Every kernel
    is deterministic, so any difference is a missing stream dependency or an allocator reuse hazard - which only long-running kernels at this size would
    expose.  8 x 3000 frames (configs[4]'s length): there the second half's recurrence runs WHILE the first half's is still going (89 ms each).

    Rounds 4-5: with the last half's decoder forward and head overlapping encoder-backward kernels this test failed in 1 run of 10-30 (last-bit
    differences of 17-221 parameters).  No dependency was missing: layernorm_bwd_kernel held `v_pk_add_f32 ... op_sel:[0,1]` (made by the compiler's SLP
    vectoriser), which misread its broadcast operand in lanes 48..63 of one wave while other queues' kernels overlapped the launch; the library is built without
    that instruction form (tests/test_build_cpu.py guards it) and the full overlap is back: 0 of 200 + 0 of 100 pairs (tools/repro_loop.py, hash_bisect.py).
    On a failure the message names the parameters that differ."""
    import bench
    from a2f_amd.trainer import TrainStep
    audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, torch.device(DEV))
    finals = []
    for _run in range(2):
        torch.manual_seed(1234)
        m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12).to(DEV)
        with torch.no_grad():
            m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
        m.train()
        step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
        assert step.pipeline
        step.pipeline_min_seqs, step.pipeline_min_frames = 1, 1   # force the two-stream form (default: from 600 frames per sequence, 28 800 per step)
        torch.manual_seed(99)
        for _ in range(2):
            out = step(audio, one_hot, gt, templ)
        torch.cuda.synchronize()
        finals.append((step.opt.flat_p.clone(), float(out["loss"].detach())))
        spans = {name: step.opt._offsets[id(p)] for name, p in m.named_parameters() if id(p) in step.opt._offsets}
        step.close()
        del step, m
    assert finals[0][1] == finals[1][1]
    if not torch.equal(finals[0][0], finals[1][0]):
        d = (finals[0][0] - finals[1][0]).abs()
        bad = [(name, float(d[o:o + n].max())) for name, (o, n) in spans.items() if float(d[o:o + n].max()) > 0]
        raise AssertionError(f"{len(bad)} of {len(spans)} parameters differ between two runs from the same state: " + ", ".join(f"{k} {v:.1e}" for k, v in bad[-12:]))


def test_faceformer_pipelined_step_under_the_bucketed_exchange_equals_the_plain_step(monkeypatch):
    """The data-parallel form of the pipelined step on the one GPU of the box: the bucketed exchange forced on (A2F_DP_FORCE_COLLECTIVES: hooks, launch
    order and the flat-buffer flow run; without a process group the all-reduce itself is skipped).  The first half's gradients are taken as autograd
    hands them out and moved into the flat buffer right before the last backward pass, which accumulates in place where the hooks reduce
    (trainer.TrainStep._begin_exchange).  Train mode with LayerDrop: parameters after two steps BITWISE equal to the step without the exchange, every
    bucket launched once in descending order."""
    import bench
    from a2f_amd.trainer import TrainStep
    S, T = 6, 90
    audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, torch.device(DEV))
    finals = []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("A2F_DP_FORCE_COLLECTIVES", "1")
        torch.manual_seed(1234)
        m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12).to(DEV)
        with torch.no_grad():
            m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
        m.train()
        m.audio_encoder.config.layerdrop = 0.3                          # some layers receive no gradient in one half or in both
        step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
        assert (step.buckets is not None) == forced
        step.pipeline_min_seqs, step.pipeline_min_frames = 0, 1         # the two-stream form at this small size
        assert step._pipelined(audio)
        torch.manual_seed(99)
        for _ in range(2):
            out = step(audio, one_hot, gt, templ)
            if forced:
                assert step._mid_collect
                assert step.buckets.launch_log == list(range(len(step.buckets.bounds) - 1, -1, -1))
        torch.cuda.synchronize()
        finals.append((step.opt.flat_p.clone(), float(out["loss"].detach())))
        step.close()
        del step, m
    assert finals[0][1] == finals[1][1]
    assert torch.equal(finals[0][0], finals[1][0])


@pytest.mark.parametrize("S,T,steps", [(6, 90, 200), (48, 600, 4)])
def test_faceformer_step_beside_collective_kernels_is_bitwise_the_plain_step(S, T, steps, monkeypatch):
    """The data-parallel step with KERNELS of a communication stream in flight beside the backward (what RCCL's all-reduce kernels are on a multi-GPU node; a
    one-GPU box cannot hold two RCCL ranks and a one-rank all-reduce launches nothing, so FlatAdam.collective injects the stand-in): every bucket's exchange is a
    copy of the gradient range out and back on its own stream, ordered like a collective library orders it (the communication stream waits for the launching
    stream; `wait()` makes the launching stream wait for it).  Summing over one rank is the identity, so the parameters after `steps` train-mode steps (LayerDrop:
    some buckets carry no gradient) must be BITWISE those of the plain step - at the small size for 200 steps, at bench size (two half-batches on two streams + the
    weight-gradient lane + the communication stream: four queues) for 4."""
    import bench
    from a2f_amd.trainer import TrainStep
    audio, one_hot, gt, templ = bench.make_ff_batch(S, T, 0, torch.device(DEV))
    comm = torch.cuda.Stream()
    launched = []

    class Work:
        def wait(self):
            torch.cuda.current_stream().wait_stream(comm)

    def collective(t, async_op):
        cur = torch.cuda.current_stream()
        comm.wait_stream(cur)
        with torch.cuda.stream(comm):
            tmp = torch.empty_like(t)
            for _ in range(2):                                          # the bytes a ring all-reduce moves, twice over: kernels that stay in flight for a while
                tmp.copy_(t)
                t.copy_(tmp)
        launched.append(t.numel())
        w = Work()
        if not async_op:
            w.wait()
            return None
        return w

    finals = []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("A2F_DP_FORCE_COLLECTIVES", "1")
        torch.manual_seed(1234)
        m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12).to(DEV)
        with torch.no_grad():
            m.vertice_map_r.weight.normal_(0, 0.1); m.vertice_map.weight.normal_(0, 0.001)
        m.train()
        m.audio_encoder.config.layerdrop = 0.3
        step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4)
        assert (step.buckets is not None) == forced
        if forced:
            step.opt.collective = collective
        step.pipeline_min_seqs, step.pipeline_min_frames = 0, 1
        assert step._pipelined(audio)
        torch.manual_seed(99)
        for _ in range(steps):
            out = step(audio, one_hot, gt, templ)
        torch.cuda.synchronize()
        finals.append((step.opt.flat_p.clone(), float(out["loss"].detach())))
        if forced:
            assert len(launched) == steps * len(step.buckets.bounds) and sum(launched) == steps * step.opt.numel     # every bucket, every step, the whole buffer
        step.close()
        del step, m
    assert finals[0][1] == finals[1][1]
    assert torch.equal(finals[0][0], finals[1][0])


def test_keep_bits_with_a_device_resident_seed_is_the_same_mask():
    """a2f_keep_bits_dev / a2f_keep_bits_attn_dev (seed read from device memory at run time: hipGraph-captured steps) against the by-value entry points."""
    be = ops.backend()
    seed = 0x1234_5678_9ABC_DEF1 & (2 ** 62 - 1)
    sd = torch.tensor([seed], dtype=torch.int64, device=DEV)
    a, b = torch.zeros(77, 24, dtype=torch.int32, device=DEV), torch.zeros(77, 24, dtype=torch.int32, device=DEV)
    be.keep_bits(seed, 5, 77, 768, 0.1, a)
    be.keep_bits(sd, 5, 77, 768, 0.1, b)
    assert torch.equal(a, b) and int(a.abs().sum()) != 0
    T, SH = 70, 6
    TW = be.attn_keep_words(T)
    q1, k1 = torch.zeros(SH * T, TW, dtype=torch.int32, device=DEV), torch.zeros(SH, TW, 32 * TW, dtype=torch.int32, device=DEV)
    q2, k2 = torch.zeros_like(q1), torch.zeros_like(k1)
    be.keep_bits_attn(seed, 9, SH, T, 0.1, q1, k1)
    be.keep_bits_attn(sd, 9, SH, T, 0.1, q2, k2)
    assert torch.equal(q1, q2) and torch.equal(k1, k2)


def test_faceformer_batch1_step_as_one_hipgraph_equals_the_eager_step():
    """The reference's own step (train.py:33: ONE sequence per optimizer step) captured as one hipGraph - device-resident mask seeds, LayerDrop as a
    device-side select, SpecAugment as a select - against the eager step with the same torch seeds: identical parameters after three train-mode steps, and
    the replays draw fresh masks (the loss changes from step to step although the inputs do not)."""
    from a2f_amd.trainer import TrainStep
    T = 48
    esd = W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0})
    dsd = W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS)
    gen = torch.Generator().manual_seed(48)
    audio = (torch.randn(1, T * 16000 // 60 + 8, generator=gen) * 0.1).to(DEV)
    oh = torch.zeros(1, 12, device=DEV); oh[0, 3] = 1
    template = (torch.randn(1, 5023, 3, generator=gen) * 5).to(DEV)
    gt = template.view(1, 1, 5023, 3) + (torch.randn(1, T, 5023, 3, generator=gen) * 0.5).to(DEV)
    finals, losses = {}, {}
    for mode in ("eager", "graph"):
        m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12)
        m.audio_encoder.load_state_dict(esd)
        m.load_state_dict(dsd, strict=False)
        m.audio_encoder.config.layerdrop = 0.3                       # several dropped layers in three steps
        m.to(DEV).train()
        step = TrainStep(m, a2f_amd.FaceFormerLoss(), lr=1e-4, use_graph=(mode == "graph"))
        torch.manual_seed(77)
        ls = []
        for _ in range(5):                                           # graph mode: two eager warm-ups, the capture step, two replays
            out = step(audio, oh, gt, template)
            ls.append(float(out["loss"].detach()))
            # the device-resident random state belongs to the step object and is visible to the model only inside its warm-up / capture: an eager forward
            # of the same model between (or after) graph steps draws on the host as ever
            assert m.audio_encoder.static_rand is None
        torch.cuda.synchronize()
        if mode == "graph":
            assert step.use_graph and step._graph is not None, "the capture failed"
            step.pipeline_min_seqs, step.pipeline_min_frames = 1, 1
            assert not step._pipelined(torch.zeros(2, audio.shape[1], device=DEV))      # a captured step never takes the two-stream form
        finals[mode], losses[mode] = step.opt.flat_p.clone(), ls
        step.close()
        del step, m
    assert len(set(losses["graph"])) == 5                            # fresh masks every replay
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(losses["eager"], losses["graph"])), (losses["eager"], losses["graph"])
    assert rel_l2(finals["graph"], finals["eager"]) < 1e-6


def test_voca_step_without_the_prediction_store_is_the_same_step():
    """Voca.store_pred = False (the labelled bench variant: a2f_head_fused with pred == NULL) - the loss and every gradient are bit for bit those of the
    API-preserving step; only the prediction is not materialised."""
    torch.manual_seed(5)
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12).to(DEV)
    B = 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 29, 16, generator=g).to(DEV)
    oh = torch.zeros(B, 12, device=DEV); oh[:, 2] = 1
    t = (torch.randn(B, 5023, 3, generator=g) * 5).to(DEV)
    gt = t + (torch.randn(B, 5023, 3, generator=g) * 0.5).to(DEV)
    from a2f_amd.trainer import FlatAdam
    opt = FlatAdam(m.parameters(), lr=1e-4)
    outs = []
    for sp in (True, False):
        m.store_pred = sp
        opt.zero_grad()
        pred, out3 = m.train_step_direct(x, oh, t, gt)
        torch.cuda.synchronize()
        assert (pred is None) == (not sp)
        outs.append((out3.clone(), opt.flat_g.clone()))
    m.store_pred = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
