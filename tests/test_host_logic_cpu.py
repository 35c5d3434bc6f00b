"""CPU: host-side composition logic of the drop-in modules, run against an EMULATOR of the C ABI.

This does NOT test the HIP kernels (tests/test_gpu_*.py do, on the MI355X box).  It proves that the
launch sequences - implicit-GEMM window strides, padding masks, gradient plumbing, state-dict layout -
reproduce the reference goldens when every a2f_* entry point honours the contract of include/a2f_hip.h.
"""
import pytest
import torch

import a2f_amd
from a2f_amd import ops
from emu_backend import EmuBackend
from golden_util import load, packed_error, rel_l2
from oracle import weights as W

NV = 15069
TOL = 1e-4


@pytest.fixture()
def emu():
    old = ops.set_backend(EmuBackend())
    yield
    ops.set_backend(old)


def test_voca_module_matches_reference_golden(emu):
    g = load("g1_voca")
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    assert W.shapes_of(m) == W.voca_shapes(NV)                    # same state-dict keys/shapes as the reference
    m.load_state_dict(W.formula_state_dict(W.voca_shapes(NV), gains={"decoder.4": 4.0}))
    B = 4
    x = W.formula_input("g1.x", (B, 29, 16))
    one_hot = torch.from_numpy(g["one_hot"])
    template = W.formula_input("g1.template", (B, 5023, 3), 5.0)
    gt = template + W.formula_input("g1.gt", (B, 5023, 3), 0.5)
    pred = m(x, one_hot, template)
    assert pred.shape == (B, 5023, 3)
    assert packed_error(g, "pred", pred) < 1e-6
    loss = a2f_amd.VocaLoss()(pred, gt)
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g["loss::" + k])) <= 1e-5 * abs(float(g["loss::" + k]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert packed_error(g, "grad::" + k, p.grad) < TOL, k


def test_voca_seeded_init_matches_torch_modules():
    """Construction order mirrors the reference so torch.manual_seed gives identical initial weights."""
    torch.manual_seed(123)
    m = a2f_amd.Voca(n_verts=33, n_onehot=12)
    torch.manual_seed(123)
    import torch.nn as nn
    convs = [nn.Conv2d(ci, co, (3, 1), (2, 1), (1, 0)) for ci, co in ((37, 32), (32, 32), (32, 64), (64, 64))]
    lins = [nn.Linear(a, b) for a, b in ((72, 72), (72, 128), (128, 50), (50, 33))]
    for i, c in zip((0, 2, 4, 6), convs):
        assert torch.equal(m.time_conv[i].weight, c.weight)
    for i, l in zip((0, 1, 3, 4), lins):
        assert torch.equal(m.decoder[i].weight, l.weight) and torch.equal(m.decoder[i].bias, l.bias)


def test_loss_rejects_odd_batch(emu):
    with pytest.raises(RuntimeError):
        a2f_amd.VocaLoss()(torch.zeros(3, 6), torch.zeros(3, 6))


def test_faceformer_loss_drops_odd_tail(emu):
    g = load("g5_losses")
    for tag, T in (("ff_t8", 8), ("ff_t9", 9)):
        p = W.formula_input("g5.p." + tag, (1, T, 5023, 3), 3.0).requires_grad_(True)
        gt = W.formula_input("g5.g." + tag, (1, T, 5023, 3), 3.0)
        l = a2f_amd.FaceFormerLoss()(p, gt)
        l["loss"].backward()
        for k in ("loss", "rec_loss", "vel_loss"):
            assert abs(float(l[k].detach()) - float(g[f"{tag}::{k}"])) <= 1e-5 * abs(float(g[f"{tag}::{k}"]))
        assert packed_error(g, f"{tag}::grad", p.grad) < 1e-5


# conv biases that feed a training-mode BatchNorm directly: their true gradient is zero (reference shows ~1e-11 noise)
BN_CANCELLED_BIASES = {f"analysis_net.{3 * l}.bias" for l in range(5)} | {f"articulation_net.{i}.bias" for i in (0, 3, 6)}


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_audio2mesh_module_matches_reference_golden(emu, mode):
    g = load("g2_audio2mesh")
    m = a2f_amd.Audio2Mesh(n_verts=NV, n_onehot=12)
    assert W.shapes_of(m) == W.audio2mesh_shapes(NV, 12)
    m.load_state_dict(W.formula_state_dict(W.audio2mesh_shapes(NV, 12), gains={"output_net.4": 4.0}))
    m.train(mode == "train")
    B = 4
    x = W.formula_input("g2.x", (B, 52, 32))
    one_hot = torch.from_numpy(g["one_hot"])
    template = W.formula_input("g2.template", (B, 5023, 3), 5.0)
    gt = template + W.formula_input("g2.gt", (B, 5023, 3), 0.5)
    pred = m(x, one_hot, template)
    assert packed_error(g, f"{mode}::pred", pred) < 1e-5
    loss = a2f_amd.VocaLoss()(pred, gt)
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g[f"{mode}::loss::{k}"])) <= 1e-5 * abs(float(g[f"{mode}::loss::{k}"]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        if mode == "train" and k in BN_CANCELLED_BIASES:
            wg = dict(m.named_parameters())[k.replace(".bias", ".weight")].grad
            assert float(p.grad.norm()) < 1e-3 * float(wg.norm()), k       # exact gradient is 0: batch-stat BN removes the conv bias
            continue
        assert packed_error(g, f"{mode}::grad::{k}", p.grad) < 5e-4, k
    if mode == "train":
        for k, b in m.named_buffers():
            if b.is_floating_point():
                assert packed_error(g, "train::buffer::" + k, b) < 1e-5, k
            else:
                assert int(b) == 1, k


def test_voca_fused_step_matches_unfused(emu):
    g = load("g1_voca")
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.voca_shapes(NV), gains={"decoder.4": 4.0}))
    B = 4
    x = W.formula_input("g1.x", (B, 29, 16))
    one_hot = torch.from_numpy(g["one_hot"])
    template = W.formula_input("g1.template", (B, 5023, 3), 5.0)
    gt = template + W.formula_input("g1.gt", (B, 5023, 3), 0.5)
    pred, loss = m.forward_with_loss(x, one_hot, template, gt)
    assert packed_error(g, "pred", pred) < 1e-6
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g["loss::" + k])) <= 1e-5 * abs(float(g["loss::" + k]))
    (loss["loss"] * 1.0).backward()
    for k, p in m.named_parameters():
        assert packed_error(g, "grad::" + k, p.grad) < TOL, k


FF_GAINS = {"vertice_map_r.weight": 4.0, "vertice_map.weight": 0.05, "obj_vector": 2.0, "in_proj_weight": 1.5}


def _faceformer(stub=True):
    m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12, audio_encoder=torch.nn.Identity() if stub else None)
    sd = W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k == "PPE.pe" or k.startswith("audio_encoder.") for k in missing), (missing, unexpected)
    return m.eval(), sd              # the deterministic function (train mode: tests/test_train_regularisers_cpu.py)


@pytest.mark.parametrize("T", [7, 61])
def test_faceformer_decoder_module_matches_reference_golden(emu, T):
    g = load("g3_faceformer_decoder")
    m, sd = _faceformer()
    pre = f"T{T}::"
    enc = W.formula_input(f"g3.enc{T}", (1, T, 768)).requires_grad_(True)
    audio = torch.zeros(1, T * 16000 // 60 + 8)
    one_hot = torch.from_numpy(g[pre + "one_hot"])
    template = W.formula_input(f"g3.template{T}", (1, 5023, 3), 5.0)
    gt = template.view(1, 1, 5023, 3) + W.formula_input(f"g3.gt{T}", (1, T, 5023, 3), 0.5)
    pred = m(audio, one_hot, template, encoder_states=enc)
    assert pred.shape == (1, T, 5023, 3)
    assert packed_error(g, pre + "pred", pred) < 1e-5
    loss = a2f_amd.FaceFormerLoss()(pred, gt)
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g[pre + "loss::" + k])) <= 1e-4 * abs(float(g[pre + "loss::" + k]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        if (pre + "nograd::" + k) in g:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k      # cross-attention q/k projections (identity 3)
            continue
        assert p.grad is not None, k
        assert packed_error(g, pre + "grad::" + k, p.grad) < TOL, k
    assert packed_error(g, pre + "grad::enc", enc.grad) < TOL


def test_faceformer_fused_loss_path_odd_T(emu):
    g = load("g3_faceformer_decoder")
    T = 7
    m, _ = _faceformer()
    pre = f"T{T}::"
    enc = W.formula_input(f"g3.enc{T}", (1, T, 768))
    audio = torch.zeros(1, T * 16000 // 60 + 8)
    one_hot = torch.from_numpy(g[pre + "one_hot"])
    template = W.formula_input(f"g3.template{T}", (1, 5023, 3), 5.0)
    gt = template.view(1, 1, 5023, 3) + W.formula_input(f"g3.gt{T}", (1, T, 5023, 3), 0.5)
    pred, loss = m.forward_with_loss(audio, one_hot, template, gt, encoder_states=enc)
    assert packed_error(g, pre + "pred", pred) < 1e-5
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g[pre + "loss::" + k])) <= 1e-4 * abs(float(g[pre + "loss::" + k]))
    loss["loss"].backward()
    for k, p in m.named_parameters():
        if (pre + "nograd::" + k) in g:
            continue
        assert packed_error(g, pre + "grad::" + k, p.grad) < TOL, k


def test_wav2vec2_module_matches_hf_golden():
    """The self-contained encoder module (HF-compatible keys), evaluated by the test-side torch-ops restatement, reproduces HF
    Wav2Vec2Model outputs (G6): pins the module's parameter layout and the restatement the emulator tests compare against."""
    import torch_ops_encoder
    from a2f_amd.model.wav2vec import Wav2Vec2Model, normalize_waveform
    g = load("g6_wav2vec2")
    m = Wav2Vec2Model().eval()
    m.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0}))
    T = 60
    audio = W.formula_input(f"g6.audio{T}", (1, T * 16000 // 60 + 3), 0.1) + 0.01
    with torch.no_grad():
        out = torch_ops_encoder.forward(m, normalize_waveform(audio), "vocaset", frame_num=T).last_hidden_state
    assert packed_error(g, f"T{T}::last_hidden_state", out) < 2e-5


def test_wav2vec2_module_loads_both_weight_norm_key_styles():
    """The positional convolution's weight norm: checkpoints written through the old nn.utils.weight_norm (the published facebook/wav2vec2-base-960h weights,
    Lightning checkpoints of older torch / transformers) hold `conv.weight_g` / `conv.weight_v`, new ones `conv.parametrizations.weight.original0 / 1`.  Both load
    (torch's own weight_norm parametrization accepts both through the same kind of pre-hook) and give the same effective weight as torch's."""
    from a2f_amd.model.wav2vec import Wav2Vec2Model
    m = Wav2Vec2Model(layers=1).eval()
    new = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    pre = "encoder.pos_conv_embed.conv."
    assert pre + "parametrizations.weight.original0" in new and pre + "parametrizations.weight.original1" in new
    m.load_state_dict(new)
    w_new = m.encoder.pos_conv_embed.conv.weight.detach().clone()
    legacy = dict(new)
    legacy[pre + "weight_g"] = legacy.pop(pre + "parametrizations.weight.original0")
    legacy[pre + "weight_v"] = legacy.pop(pre + "parametrizations.weight.original1")
    m2 = Wav2Vec2Model(layers=1).eval()
    m2.load_state_dict(legacy)                                      # strict: no missing / unexpected keys
    assert torch.equal(m2.encoder.pos_conv_embed.conv.weight, w_new)
    ref = torch.nn.utils.parametrizations.weight_norm(torch.nn.Conv1d(768, 768, 128, padding=64, groups=16), name="weight", dim=2)
    ref.load_state_dict({k[len(pre):]: v for k, v in legacy.items() if k.startswith(pre)})
    assert torch.allclose(ref.weight, w_new, rtol=1e-6, atol=1e-7)
    # a Faceformer-level dict (keys under audio_encoder.) goes through the same hook
    import a2f_amd
    ff = a2f_amd.Faceformer(n_verts=15, n_onehot=12, audio_encoder=Wav2Vec2Model(layers=1))
    sd = ff.state_dict()
    p2 = "audio_encoder." + pre
    sd[p2 + "weight_g"], sd[p2 + "weight_v"] = sd.pop(p2 + "parametrizations.weight.original0"), sd.pop(p2 + "parametrizations.weight.original1")
    ff.load_state_dict(sd)


def test_wav2vec2_module_has_no_cpu_path():
    """The product module has one forward - the HIP path: with the real backend, off-device tensors fail loudly."""
    from a2f_amd.model.wav2vec import Wav2Vec2Model
    m = Wav2Vec2Model(layers=1).eval()
    with pytest.raises(Exception) as ei:
        m(torch.zeros(1, 4000), "vocaset", frame_num=10)
    assert "HIP" in str(ei.value) or "liba2f_hip" in str(ei.value) or "a2f" in str(ei.value).lower()


def test_faceformer_end_to_end_golden_g8(emu):
    g = load("g8_faceformer_e2e")
    m, _ = _faceformer(stub=False)
    m.audio_encoder.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0}))
    m.eval()
    T = 30
    audio = W.formula_input("g8.audio", (1, T * 16000 // 60 + 5), 0.1)
    one_hot = torch.from_numpy(g["one_hot"])
    template = W.formula_input("g8.template", (1, 5023, 3), 5.0)
    with torch.no_grad():
        pred = m(audio, one_hot, template)
    assert packed_error(g, "pred", pred) < 1e-5


def test_voca_accepts_per_frame_features(emu):
    """SURVEY §8(f) N4 host logic: Voca fed with (S,T,29) per-frame features must equal Voca fed with the materialised
    16-frame zero-padded windows (a2f_amd.data.feature_windows) - prediction, loss and every gradient - and the oracle."""
    from a2f_amd.data import feature_windows
    from oracle import voca as o_voca
    S, T = 2, 6
    feats = W.formula_input("n4.feats", (S, T, 29))
    B = S * T
    one_hot = torch.zeros(B, 12); one_hot[torch.arange(B), torch.arange(B) % 8] = 1.0
    template = W.formula_input("n4.template", (B, 5023, 3), 5.0)
    gt = template + W.formula_input("n4.gt", (B, 5023, 3), 0.5)
    sd = W.formula_state_dict(W.voca_shapes(NV), gains={"decoder.4": 4.0})
    outs = []
    for x in (feature_windows(feats), feats):
        m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
        m.load_state_dict(sd)
        pred = m(x, one_hot, template)
        loss = a2f_amd.VocaLoss()(pred, gt)
        loss["loss"].backward()
        outs.append((pred.detach(), float(loss["loss"]), [p.grad.clone() for p in m.parameters()]))
    assert outs[0][0].shape == (B, 5023, 3) and torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)
    ref = o_voca.voca_forward({k: v for k, v in sd.items()}, feature_windows(feats), one_hot, template)
    assert rel_l2(outs[1][0] - template, ref - template) < 1e-5
    # window semantics: frame t of a sequence sees rows t-8 .. t+7 of ITS sequence, zeros outside
    w = feature_windows(feats)
    assert torch.equal(w[0, :, 8], feats[0, 0]) and torch.all(w[0, :, :8] == 0) and torch.equal(w[T, :, 8], feats[1, 0]) and torch.all(w[T, :, :8] == 0)


@pytest.mark.parametrize("images", [False, True])
def test_hip_encoder_composition_matches_oracle(emu, images, monkeypatch):
    """The hand-written encoder path (encoder_ops: a2f_mm windows / batches / epilogues, flash-attention and LayerNorm contracts,
    fixed backward launch sequence) composed over the float64 emulator of the C ABI against the oracle's autograd: forward,
    d/d(waveform) and every parameter gradient of the first layers (two sequences, T = 12).  images: the Transformer layers on split-32 image
    operands (what the product runs from 2048 rows: producer-side images, image GEMM epilogues, masked LayerNorm backward) at this small size."""
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0 if images else 1 << 30)
    from a2f_amd.model.wav2vec import Wav2Vec2Model
    from oracle import wav2vec2 as o_w2v
    m = Wav2Vec2Model(layers=2).eval()                 # the deterministic function (train mode: tests/test_train_regularisers_cpu.py)
    sd = {k: v for k, v in W.formula_state_dict(W.wav2vec2_shapes(layers=2), gains={"original0": 40.0}).items()}
    m.load_state_dict(sd)
    T = 12
    wave = o_w2v.normalize_waveform(W.formula_input("enc.audio", (2, T * 16000 // 60 + 5), 0.1))
    wg = wave.clone().requires_grad_(True)
    h = m._forward_hip(wg, T)
    E = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    wo = wave.clone().requires_grad_(True)
    ref = torch.cat([o_w2v.wav2vec2_forward(E, wo[i:i + 1], T, n_layers=2) for i in range(2)], 0)
    assert ("mm", 24, 3072, 768, 1) in ops.backend().calls
    # (the positional convolution's kernels take their weight as an image in either setting: 768 rows of 128 taps x 48 channels)
    assert any(c[0] == "split_img" and (c[1], c[2]) != (768, 6144) for c in ops.backend().calls) == images
    assert any(c[0] == "posconv_nt" for c in ops.backend().calls)
    # images: every activation between two products is rounded to 16 significant bits (hi + lo) instead of 24; 13 such stages here
    assert rel_l2(h, ref) < (2e-5 if images else 1e-5)
    gout = W.formula_input("enc.gout", tuple(ref.shape))
    h.backward(gout)
    ref.backward(gout)
    assert rel_l2(wg.grad, wo.grad) < 1e-4
    P = dict(m.named_parameters())
    for k, v in E.items():
        if v.grad is None or k.endswith("k_proj.bias"):          # d/d(k bias) is identically zero in exact arithmetic (softmax shift invariance)
            continue
        assert rel_l2(P[k].grad, v.grad) < 1e-4, k


def test_flat_adam_release_and_collect_grads(emu):
    """trainer.FlatAdam.release_grads / collect_grads: gradients autograd hands over (fresh tensors, views, a parameter that received none - a
    layer LayerDrop skipped) end up in the flat buffer exactly as zero-fill + accumulation would have left them, and ``p.grad`` is the flat view
    again; a second backward before the collect accumulates in place."""
    from a2f_amd.trainer import FlatAdam
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    opt = FlatAdam(ps, lr=1e-3)
    views = [p.grad for p in ps]
    x = torch.randn(4, 5)

    def loss():
        return (x @ ps[0]).pow(2).sum() + (ps[1][:3] * 2.0).sum()           # ps[2] takes no part

    opt.zero_grad()
    loss().backward(); loss().backward()
    want = opt.flat_g.clone()
    opt.flat_g.fill_(7.0)                                                    # stale values must not survive
    opt.release_grads()
    assert all(p.grad is None for p in ps)
    loss().backward(); loss().backward()
    assert ps[2].grad is None
    opt.collect_grads()
    assert all(p.grad is v for p, v in zip(ps, views))
    o2, n2 = opt._offsets[id(ps[2])][0], ps[2].numel()
    assert torch.equal(opt.flat_g[o2:o2 + n2], torch.zeros(n2))
    for p in ps[:2]:
        o, n = opt._offsets[id(p)][0], p.numel()
        assert torch.equal(opt.flat_g[o:o + n], want[o:o + n])
    # the pipelined step's form: the first pass set aside (stash_grads), the last pass handed out as it is, ONE multi-tensor add in collect_grads - the same
    # sums, bit for bit; a parameter only the first pass reached (ps[1] here) and one no pass reached (ps[2]) included
    def loss_b():
        return (x @ ps[0]).pow(2).sum() * 0.5
    opt.zero_grad()
    loss().backward(); loss_b().backward()
    want = opt.flat_g.clone()
    opt.flat_g.fill_(7.0)
    opt.release_grads()
    loss().backward()
    opt.stash_grads()
    assert all(p.grad is None for p in ps)
    loss_b().backward()
    assert ps[1].grad is None
    opt.collect_grads()
    assert all(p.grad is v for p, v in zip(ps, views)) and not opt._stash
    for p in ps:                                                             # (segments are padded to 16 bytes: the padding is nobody's)
        o, n = opt._offsets[id(p)][0], p.numel()
        assert torch.equal(opt.flat_g[o:o + n], want[o:o + n])


def test_bucketed_allreduce_refuses_a_model_that_breaks_its_order_assumption():
    """trainer.BucketedAllReduce launches buckets in reverse parameter order as the backward passes them.  A parameter that is registered early but used late
    (filed under the first bucket, its gradient arrives first) would send every later bucket out before its gradients exist: the hook must fail loudly."""
    import torch.nn as nn
    from a2f_amd.trainer import BucketedAllReduce, FlatAdam

    class Net(nn.Module):
        def __init__(self, late_front):
            super().__init__()
            self.front = nn.Linear(4, 4)
            self.layers = nn.ModuleList(nn.Linear(4, 4) for _ in range(3))
            self.late_front = late_front

        def forward(self, x):
            if not self.late_front:
                x = self.front(x)
            for l in self.layers:
                x = torch.tanh(l(x))
            return self.front(x) if self.late_front else x

    key = lambda name: int(name.split(".")[1]) + 1 if name.startswith("layers.") else 0
    for late in (False, True):
        torch.manual_seed(0)
        m = Net(late)
        opt = FlatAdam(m.parameters(), lr=1e-3)
        br = BucketedAllReduce(opt, list(m.named_parameters()), key)
        opt.zero_grad()
        loss = m(torch.randn(5, 4)).square().sum()
        br.begin()
        if late:
            with pytest.raises(RuntimeError, match="arrived after that bucket's all-reduce was started"):
                loss.backward()
        else:
            loss.backward()
            assert br.finish() and br.launch_log == [3, 2, 1, 0]
        br.close()
