"""CPU: config.yaml surface and registries mirror the reference harness (lightning_model.py:20-73, config.yaml:1-12)."""
import os
import tempfile

import pytest

import a2f_amd
from a2f_amd.registry import ExpConfig, build_from_config, get_loss_fn, get_model

REFERENCE_CONFIG = """batch_size: 128
modelname: "audio2mesh"
vertex_count: 15069 #5023 * 3
one_hot_size: 12
split_frame: True
percision: "16-mixed"
lr: 1e-4
feature_extractor: "mfcc"
sample_rate: 22000
n_feature: 32
out_dim: 52
win_length: 440 #220*2
"""


import pytest


@pytest.fixture
def emu():
    """The package computes through ops.backend(): on this GPU-less box the float64 emulator of the C ABI stands in for liba2f_hip.so."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from a2f_amd import ops
    from emu_backend import EmuBackend
    old = ops.set_backend(EmuBackend())
    yield
    ops.set_backend(old)


def test_expconfig_parses_reference_yaml_keys():
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "config.yaml")
        open(p, "w").write(REFERENCE_CONFIG)
        cfg = ExpConfig.from_yaml(p)
    assert cfg.modelname == "audio2mesh" and cfg.vertex_count == 15069 and cfg.percision == "16-mixed" and cfg.lr == 1e-4
    assert cfg.hop_length is None and cfg.loss is None
    assert cfg.name() == "audio2mesh_mfcc_0.0001_None_16-mixed"
    model, loss = build_from_config(cfg)
    assert isinstance(model, a2f_amd.Audio2Mesh) and model.n_verts == 15069 and model.n_onehot == 12
    assert isinstance(loss, a2f_amd.VocaLoss)


def test_percision_key_maps_to_encoder_autocast():
    """`percision` is Lightning's precision flag (lightning_model.py / config.yaml:6): the mixed settings autocast the
    torch-op wav2vec2 encoder of FaceFormer, the 32-bit settings leave it alone; the HIP-side models have no such switch."""
    import torch
    from a2f_amd.registry import mixed_precision_dtype
    with pytest.warns(UserWarning, match="GradScaler"):            # no loss scaling in TrainStep: fp16 is run as bf16, loudly
        assert mixed_precision_dtype("16-mixed") is torch.bfloat16
    assert mixed_precision_dtype("bf16-mixed") is torch.bfloat16
    assert mixed_precision_dtype("32-true") is None and mixed_precision_dtype("32") is None
    assert a2f_amd.Faceformer.encoder_autocast is None and not hasattr(a2f_amd.Voca, "encoder_autocast")


def test_registries():
    assert get_model("voca") is a2f_amd.Voca and get_model("faceformer") is a2f_amd.Faceformer
    assert isinstance(get_loss_fn("faceformer"), a2f_amd.FaceFormerLoss) and isinstance(get_loss_fn("voca"), a2f_amd.VocaLoss)
    with pytest.raises(KeyError):
        get_model("song2face")


def test_resampler_properties():
    """N3: 22 kHz -> 16 kHz sinc/Hann resampler restated from torchaudio's published algorithm (parity unpinned: torchaudio
    is absent) - length, DC gain and an in-band tone are checked instead."""
    import math
    import torch
    from a2f_amd.data import resample_sinc_hann
    L = 22000
    x = torch.ones(2, L)
    y = resample_sinc_hann(x, 22000, 16000)
    assert y.shape == (2, math.ceil(16000 * L / 22000))
    assert torch.allclose(y[:, 200:-200], torch.ones_like(y[:, 200:-200]), atol=2e-3)          # unit DC gain away from the zero-padded ends
    t = torch.arange(L) / 22000.0
    tone = torch.sin(2 * math.pi * 440.0 * t)
    r = resample_sinc_hann(tone, 22000, 16000)
    t2 = torch.arange(r.shape[-1]) / 16000.0
    ref = torch.sin(2 * math.pi * 440.0 * t2)
    assert (r[200:-200] - ref[200:-200]).abs().max() < 5e-3
    assert resample_sinc_hann(tone, 16000, 16000) is tone


def test_mfcc_extractor_structure(emu):
    """N2 (parity unpinned: torchaudio absent): the MFCC front end restated with torch ops - output shape of the
    config.yaml setting (0.52 s @ 22 kHz -> (B, 52, 32)), orthonormal DCT, triangular mel filters, top_db floor."""
    import torch
    from a2f_amd.extractor import MFCCExtractor, create_dct_ortho, get_extractor, melscale_fbanks_htk
    ex = get_extractor("mfcc")(sample_rate=22000, n_feature=32, out_dim=52, win_length=440, hop_length=None, n_fft=1024)
    x = torch.randn(3, 11440) * 0.1
    y = ex(x)
    assert y.shape == (3, 52, 32) and torch.isfinite(y).all()
    assert ex.hop_length == 220 and ex.n_fft == 1024
    d = create_dct_ortho(128, 128)
    assert torch.allclose(d.t() @ d, torch.eye(128), atol=1e-5)                  # full DCT-II with ortho scaling is orthonormal
    fb = melscale_fbanks_htk(513, 0.0, 11000.0, 128, 22000)
    assert fb.shape == (513, 128) and (fb >= 0).all() and (fb.max(0).values <= 1.0 + 1e-6).all()
    peaks = fb.argmax(0)
    assert (peaks[1:] >= peaks[:-1]).all()                                        # centre frequencies increase
    from torch_ops_mfcc import mel_spectrogram_stft
    mel = mel_spectrogram_stft(ex, x)
    assert mel.shape[1] == 128 and mel.shape[2] == 11440 // 220 + 1               # centred frames: 53 -> resized to 52
    assert tuple(ex.mel_spectrogram(x).shape) == tuple(mel.shape)
    # a louder clip shifts every coefficient-0 by the same dB amount (log-power linearity), others unchanged
    y2 = ex(x * 10.0)
    assert torch.allclose(y2[..., 1:], y[..., 1:], atol=2e-2)


def test_build_extractor_from_config(emu):
    """The reference's own config.yaml (audio2mesh + mfcc) builds the MFCC extractor; this repo's VOCA config (features fed
    directly, feature_extractor: null) builds none."""
    import os
    import torch
    from a2f_amd.registry import ExpConfig, build_extractor
    ref_cfg = ExpConfig(batch_size=128, modelname="audio2mesh", vertex_count=15069, one_hot_size=12, split_frame=True, percision="16-mixed",
                        lr=1e-4, feature_extractor="mfcc", sample_rate=22000, n_feature=32, out_dim=52, win_length=440)
    ex = build_extractor(ref_cfg)
    assert ex is not None and ex(torch.zeros(2, 11440)).shape == (2, ref_cfg.out_dim, ref_cfg.n_feature)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert build_extractor(ExpConfig.from_yaml(os.path.join(root, "config.yaml"))) is None


def test_mfcc_top_db_floor_is_batch_wide(emu):
    """AmplitudeToDB on the (batch, mel, time) spectrogram floors at the maximum over the WHOLE batch minus 80 dB (torchaudio
    packs a 3-D input as one item): a quiet clip next to a loud one is clamped harder than when it is processed alone."""
    import torch
    from a2f_amd.extractor import MFCCExtractor
    ex = MFCCExtractor(sample_rate=22000, n_feature=32, out_dim=52, win_length=440, hop_length=None, n_fft=1024)
    g = torch.Generator().manual_seed(3)
    loud = torch.randn(1, 11440, generator=g)
    quiet = torch.randn(1, 11440, generator=g) * 1e-6
    both = ex(torch.cat([loud, quiet], 0))
    alone = ex(quiet)
    assert torch.allclose(both[0], ex(loud)[0], atol=1e-4)            # the loud clip sets the floor: unchanged
    assert not torch.allclose(both[1], alone[0], atol=1e-2)           # the quiet clip is floored by the loud clip's maximum
    from torch_ops_mfcc import mel_spectrogram_stft
    mel = mel_spectrogram_stft(ex, torch.cat([loud, quiet], 0))
    db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    assert float(db[1].max()) < float(db.max()) - 80.0                # i.e. the whole quiet clip sits below the batch-wide floor


def test_wav2vec_extractor_matches_hf_golden_g12():
    """N2: Wav2VecExtractor (extractor.py:63-96) after the resampler against fixture G12 = the installed HF processor + HF
    Wav2Vec2Model + the reference's resize, formula weights: the joint (batch, time) normalisation quirk, the plain 49 Hz
    forward, transpose and bilinear resize.  The encoder inside runs the product's only path - the hand-written one - over the float64
    emulator of the C ABI (the GPU run of the same fixture: tests/test_gpu_models.py::test_wav2vec_extractor_on_gpu_g12)."""
    import numpy as np
    import torch
    from a2f_amd import ops
    from a2f_amd.extractor import Wav2VecExtractor, get_extractor
    from emu_backend import EmuBackend
    from golden_util import load, packed_error
    from oracle import weights as W
    g = load("g12_wav2vec_extractor")
    assert get_extractor("wav2vec") is Wav2VecExtractor
    for name in ("a", "b"):
        B, n, out_dim, n_feature = (int(v) for v in g[f"{name}::dims"])
        ex = Wav2VecExtractor(sample_rate=16000, n_feature=n_feature, out_dim=out_dim).eval()     # 16 kHz in: the resampler is the identity
        ex.model.load_state_dict(W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0}))
        assert not any(p.requires_grad for p in ex.model.feature_extractor.parameters())          # freeze_feature_encoder
        x = W.formula_input(f"g12.audio.{name}", (B, n), 0.1) + 0.02 * torch.arange(B).view(B, 1)
        old = ops.set_backend(EmuBackend())
        try:
            with torch.no_grad():
                y = ex(x)
        finally:
            ops.set_backend(old)
        assert tuple(y.shape) == (B, out_dim, n_feature)
        assert packed_error(g, f"{name}::features", y) < 2e-5


def test_mfcc_matrix_product_form_equals_stft_form():
    """The HIP evaluation of the MFCC front end (windowed DFT basis x frames, mel filter bank and DCT as three products of the library's
    GEMM) composed over the float64 emulator of the C ABI against the torch.stft form of the same definition: mel spectrogram and
    MFCC agree to fp32 rounding (both forms stay PARITY-UNPINNED against torchaudio, which is not in this image)."""
    import torch
    from a2f_amd import ops
    from a2f_amd.extractor import MFCCExtractor
    from emu_backend import EmuBackend
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(3, 11440, generator=gen) * 0.1
    x[1] *= 1e-3                                                             # one quiet clip: the batch-wide top_db floor becomes active
    ex = MFCCExtractor(22000, 32, 52, 440, None, 1024)
    from torch_ops_mfcc import mel_spectrogram_stft, mfcc_torch_ops
    ref_mel, ref = mel_spectrogram_stft(ex, x), mfcc_torch_ops(ex, x)
    old = ops.set_backend(EmuBackend())
    try:
        mel = ex.mel_spectrogram(x)
        assert mel.shape == ref_mel.shape and float((mel - ref_mel).norm() / ref_mel.norm()) < 1e-6
        got = ex(x)
    finally:
        ops.set_backend(old)
    assert got.shape == ref.shape == (3, 52, 32)
    assert float((torch.as_tensor(got) - ref).norm() / ref.norm()) < 1e-5
