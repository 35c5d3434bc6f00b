"""Train-mode stochastic regularisers of the wav2vec2 encoder (dropout sites, LayerDrop, SpecAugment) on the CPU: the mask
plumbing, the torch-ops restatement and the HIP-path composition (over the float64 emulator of the C ABI) against fixture G13 -
the reference's own train-mode forward / backward with every mask recorded (tests/golden/gen_golden.py g13)."""
import numpy as np
import pytest
import torch

import a2f_amd
from a2f_amd import dropout as D
from a2f_amd import ops
from emu_backend import EmuBackend, philox_keep_words, unpack_keep_bits
from golden_util import load, packed_error, rel_l2
from oracle import wav2vec2 as o_w2v
from oracle import weights as W

GRAD_KEYS = ("masked_spec_embed", "feature_projection.projection.weight", "feature_projection.projection.bias", "encoder.layers.0.attention.q_proj.weight",
             "encoder.layers.0.attention.v_proj.bias", "encoder.layers.5.feed_forward.intermediate_dense.weight",
             "encoder.layers.11.feed_forward.output_dense.weight", "encoder.layers.11.final_layer_norm.weight",
             "feature_extractor.conv_layers.0.conv.weight", "encoder.layer_norm.bias")


def recorded_masks(g):
    """fixture G13 -> the dict EncoderMasks.from_recorded takes (boolean masks in the shapes the kernels index)."""
    S, T, _n = (int(v) for v in g["dims"])
    un = lambda key, shape: np.unpackbits(g[key])[: int(np.prod(shape))].reshape(shape).astype(bool)
    rec = {"featproj": un("featproj", (S * T, 768)), "enc": un("enc", (S * T, 768)), "layer_skip": g["layer_skip"], "spec": g["spec"]}
    for l in range(12):
        if f"layer{l}.attn" not in g:
            continue
        rec[f"layer{l}.attn"] = un(f"layer{l}.attn", (S * 12, T, T))
        rec[f"layer{l}.hid1"] = un(f"layer{l}.hid1", (S * T, 768))
        rec[f"layer{l}.act"] = un(f"layer{l}.act", (S * T, 3072))
        rec[f"layer{l}.hid2"] = un(f"layer{l}.hid2", (S * T, 768))
    return rec, S, T


def g13_model_and_input(device="cpu"):
    g = load("g13_wav2vec2_train_masks")
    rec, S, T = recorded_masks(g)
    m = a2f_amd.model.wav2vec.Wav2Vec2Model()
    m.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0}))
    m.to(device).train()
    m.injected_masks = D.EncoderMasks.from_recorded(rec, device)
    n = int(g["dims"][2])
    wave = o_w2v.normalize_waveform(W.formula_input("g13.audio", (S, n), 0.1) + 0.01).to(device)
    return g, m, wave, S, T


def check_against_g13(g, m, hs, T, tol_out, tol_grad):
    assert packed_error(g, "last_hidden_state", hs) < tol_out
    (hs * W.formula_input("g13.cotangent", tuple(hs.shape), 1.0).to(hs.device)).sum().backward()
    P = dict(m.named_parameters())
    for k in GRAD_KEYS:
        assert packed_error(g, f"grad::{k}", P[k].grad) < tol_grad, k
    for k, p in P.items():                                   # LayerDrop: the skipped layers take no part in the step
        if k.startswith("encoder.layers.3.") or k.startswith("encoder.layers.8."):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


def test_spec_augment_port_matches_reference_stream():
    """compute_mask_indices consumes numpy's stream exactly as wav2vec.py:25-72 does: the mask the reference drew from
    np.random.seed(1234) inside its train-mode forward (G13) comes out of the same seed."""
    g = load("g13_wav2vec2_train_masks")
    S, T, _ = (int(v) for v in g["dims"])
    assert np.array_equal(D.compute_mask_indices((S, T), 0.05, 10, min_masks=2, rng=np.random.RandomState(1234)), g["spec"])
    np.random.seed(1234)                                     # default: numpy's global state, what the reference draws from
    assert np.array_equal(D.compute_mask_indices((S, T), 0.05, 10, min_masks=2), g["spec"])
    m = D.compute_mask_indices((5, 600), 0.05, 10, min_masks=2, rng=np.random.RandomState(3))
    assert m.shape == (5, 600) and len({int(r.sum()) for r in m}) == 1 and 10 <= int(m[0].sum()) <= 40      # every row thinned to the shortest


def test_keep_bit_layouts_round_trip():
    gen = torch.Generator().manual_seed(5)
    mask = torch.rand(7, 100, generator=gen) > 0.3
    bits = D.pack_keep_bits(mask)
    assert bits.dtype == torch.int32 and bits.shape == (7, 4) and torch.equal(D.unpack_keep_bits(bits, 100), mask)
    am = torch.rand(3, 70, 70, generator=gen) > 0.1
    q, k = D.pack_keep_bits_attn(am)
    TW = D.attn_keep_words(70)
    assert TW == 4 and q.shape == (3 * 70, TW) and k.shape == (3, TW, 32 * TW)
    assert torch.equal(D.unpack_keep_bits(q, 70).view(3, 70, 70), am)
    kk = D.unpack_keep_bits(k.transpose(1, 2).reshape(3 * 32 * TW, TW), 70).view(3, 32 * TW, 70)[:, :70]      # [sh][key][query]
    assert torch.equal(kk.transpose(1, 2), am)


def test_philox_restatement_known_answers():
    """The numpy restatement of a2f_keep_bits is Philox4x32-10 (Random123 known-answer vectors through the same round function),
    and its keep rate follows p."""
    import emu_backend
    w = philox_keep_words(0, 0, 1, 4, 0.0, 1)              # p = 0 keeps everything; checks the packing only
    assert int(w[0, 0]) == 0xF
    # counter (0, 0, 0, 0), key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8 (Random123 kat_vectors): eight 16-bit randoms, low half first:
    # e8d5 6627 c58d e169 ac4c bc57 dbd8 9b00; an element keeps iff its random >= thresh
    for thresh16, expect in ((0x6627, (1, 1, 1, 1, 1, 1, 1, 1)), (0x6628, (1, 0, 1, 1, 1, 1, 1, 1)), (0xbc58, (1, 0, 1, 1, 0, 0, 1, 0)), (0xe8d6, (0,) * 8)):
        w = philox_keep_words(0, 0, 1, 8, thresh16 / 65536.0, 1)
        assert tuple((int(w[0, 0]) >> e) & 1 for e in range(8)) == expect
    k = unpack_keep_bits(torch.from_numpy(philox_keep_words(987654321987, 5, 64, 768, 0.1, 24)), 768)
    assert abs(float(k.float().mean()) - 0.9) < 0.01


def test_torch_ops_train_mode_matches_reference_g13():
    """The torch-ops restatement of the wrapper in train mode, fed the reference's recorded masks."""
    g, m, wave, S, T = g13_model_and_input()
    import torch_ops_encoder
    hs = torch_ops_encoder.forward(m, wave, "vocaset", frame_num=T).last_hidden_state
    check_against_g13(g, m, hs, T, 2e-5, 1e-4)


@pytest.mark.parametrize("images", [False, True])
def test_hip_path_composition_train_mode_matches_reference_g13(images, monkeypatch):
    """The hand-written path's composition in train mode (dropout in the a2f_mm epilogues and the attention kernels' contract, the
    masked backward passes, LayerDrop, SpecAugment) over the float64 emulator of the C ABI against G13; images: the Transformer layers on
    split-32 image operands (masks inside the image GEMM epilogues and the LayerNorm backward)."""
    from a2f_amd import encoder_ops
    monkeypatch.setattr(encoder_ops, "IMG_MIN_ROWS", 0 if images else 1 << 30)
    g, m, wave, S, T = g13_model_and_input()
    old = ops.set_backend(EmuBackend())
    try:
        hs = m._forward_hip(wave, T)
        check_against_g13(g, m, hs, T, 2e-5, 1e-4)
    finally:
        ops.set_backend(old)


def test_drawn_masks_are_reproducible_and_eval_is_deterministic():
    """Masks drawn from torch's generator: same seed -> same step; .eval() and train_regularisers = False -> the deterministic function."""
    m = a2f_amd.model.wav2vec.Wav2Vec2Model(layers=2)
    m.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(layers=2), gains={"original0": 40.0}))
    T = 12
    wave = o_w2v.normalize_waveform(W.formula_input("drawn.audio", (2, T * 16000 // 60 + 5), 0.1))
    old = ops.set_backend(EmuBackend())
    try:
        m.train()
        torch.manual_seed(11)
        a = m._forward_hip(wave, T).detach()
        torch.manual_seed(11)
        b = m._forward_hip(wave, T).detach()
        c = m._forward_hip(wave, T).detach()
        assert torch.equal(a, b) and not torch.equal(a, c)
        m.train_regularisers = False
        d = m._forward_hip(wave, T).detach()
        m.eval()
        m.train_regularisers = True
        e = m._forward_hip(wave, T).detach()
        assert torch.equal(d, e) and rel_l2(a, e) > 1e-3
    finally:
        ops.set_backend(old)


def test_device_resident_step_randomness_equals_the_host_drawn_step():
    """dropout.StepRandomness (the random state of a hipGraph-captured step: mask seeds read from device memory, LayerDrop as a device-side select over
    layers that all run, SpecAugment as a device-side select) against the eager step that draws on the host: same torch seed -> same masks and decisions
    -> the same output and the same gradients (a dropped layer's come out as exact zeros instead of None)."""
    from a2f_amd.dropout import StepRandomness
    L = 4
    m = a2f_amd.model.wav2vec.Wav2Vec2Model(layers=L)
    m.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(layers=L), gains={"original0": 40.0}))
    m.config.layerdrop = 0.5                                   # make dropped layers the rule
    T = 14
    wave = o_w2v.normalize_waveform(W.formula_input("static.audio", (2, T * 16000 // 60 + 5), 0.1))
    gout = W.formula_input("static.gout", (2, T, 768))
    old = ops.set_backend(EmuBackend())
    try:
        m.train()
        seen_drop = False
        for seed in (3, 4, 5):
            torch.manual_seed(seed)
            for p in m.parameters():
                p.grad = None
            a = m._forward_hip(wave, T)
            a.backward(gout)
            skipped = list(m.last_masks.layer_skip)
            ga = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
            sr = m.static_rand = StepRandomness("cpu", L, 2, T)
            torch.manual_seed(seed)
            sr.refresh(m.config)
            assert [not k for k in sr.last["layer_keep"]] == skipped
            for p in m.parameters():
                p.grad = None
            b = m._forward_hip(wave, T)
            b.backward(gout)
            m.static_rand = None
            seen_drop = seen_drop or any(skipped)
            assert torch.equal(a.detach(), b.detach())
            for k, p in m.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                assert torch.equal(g, ga[k]) or rel_l2(g, ga[k]) < 1e-6, k
        assert seen_drop
    finally:
        m.static_rand = None
        ops.set_backend(old)


# ---------------------------------------------------------------------------------------------------------------------------
# FaceFormer decoder / PPE dropout with per-row masks (fixture G14)
# ---------------------------------------------------------------------------------------------------------------------------
NV = 15069
FF_GAINS = {"vertice_map_r.weight": 4.0, "vertice_map.weight": 0.05, "obj_vector": 2.0, "in_proj_weight": 1.5}


def g14_recorded(g, T):
    pre = f"T{T}::mask::"
    un = lambda key, shape: np.unpackbits(g[pre + key])[: int(np.prod(shape))].reshape(shape).astype(bool)
    cross = un("cross", (4, T, T))
    return {"ppe": un("ppe", (T, 64)), "d1": un("d1", (T, 64)), "d2": un("d2", (T, 64)), "act": un("act", (T, 128)), "d3": un("d3", (T, 64)),
            "attn": un("attn", (4, T, T)),
            # the diagonal memory mask leaves key i for frame i: only that probability's keep decision matters
            "cross_head": np.stack([cross[h, np.arange(T), np.arange(T)] for h in range(4)], axis=1)}


def g14_case(T, device="cpu"):
    g = load("g14_faceformer_decoder_train_masks")
    m = a2f_amd.Faceformer(n_verts=NV, n_onehot=12, audio_encoder=torch.nn.Identity())
    m.load_state_dict(W.formula_state_dict(W.faceformer_decoder_shapes(NV, 12, 64), gains=FF_GAINS), strict=False)
    m.to(device).train()
    m.injected_decoder_masks = D.DecoderMasks.from_recorded(g14_recorded(g, T), device)
    pre = f"T{T}::"
    enc = W.formula_input(f"g14.enc{T}", (1, T, 768)).to(device).requires_grad_(True)
    audio = torch.zeros(1, T * 16000 // 60 + 8, device=device)
    one_hot = torch.from_numpy(g[pre + "one_hot"]).to(device)
    template = W.formula_input(f"g14.template{T}", (1, 5023, 3), 5.0)
    gt = (template.view(1, 1, 5023, 3) + W.formula_input(f"g14.gt{T}", (1, T, 5023, 3), 0.5)).to(device)
    return g, m, pre, enc, audio, one_hot, template.to(device), gt


def check_against_g14(g, m, pre, enc, pred, loss, tol):
    assert packed_error(g, pre + "pred", pred) < tol
    for k in ("loss", "rec_loss", "vel_loss"):
        assert abs(float(loss[k].detach()) - float(g[pre + "loss::" + k])) <= 1e-4 * abs(float(g[pre + "loss::" + k]))
    loss["loss"].backward()
    seen = 0
    for k, p in m.named_parameters():
        if (pre + "grad::" + k) in g or (pre + "grad::" + k + "::sample") in g:
            assert packed_error(g, pre + "grad::" + k, p.grad) < tol, k
            seen += 1
    assert seen >= 20
    assert packed_error(g, pre + "grad::enc", enc.grad) < tol


@pytest.mark.parametrize("T", [7, 61])
def test_faceformer_decoder_train_mode_matches_reference_with_per_row_masks_g14(T):
    """Decoder recurrence + cross rows + head in train mode over the float64 emulator of the C ABI, fed the per-row masks fixture G14
    was generated with (the reference's O(T^3) loop with its dropout calls replaced by row-consistent masks)."""
    g, m, pre, enc, audio, one_hot, template, gt = g14_case(T)
    old = ops.set_backend(EmuBackend())
    try:
        pred = m(audio, one_hot, template, encoder_states=enc)
        check_against_g14(g, m, pre, enc, pred, a2f_amd.FaceFormerLoss()(pred, gt), 1e-4)
    finally:
        ops.set_backend(old)
