"""Audio front end in front of the Audio2Mesh path: ``MFCCExtractor``.

Reference: /root/reference/src/model/extractor.py:10-59 - a thin wrapper around ``torchaudio.transforms.MFCC`` followed by
a bilinear resize to ``out_dim`` frames; constructed at lightning_model.py:86-93 as
``MFCCExtractor(sample_rate=22000, n_feature=32, out_dim=52, win_length=440, hop_length=None, n_fft=1024)`` and applied
to the 0.52 s audio window of a frame (``feature = extractor(x).detach()``, lightning_model.py:115: no gradient flows
into it, so it is a pure pre-processing step upstream of the hot path).

torchaudio is NOT in this image and the reference pins no version, so the arithmetic below is a restatement of torchaudio's
published defaults - **PARITY UNPINNED**: there is no torchaudio build, reference test or fixture here to check it against; tests
pin structural properties only.  ONE evaluation in the product: the STFT / mel filter bank / DCT as three products on the library's
exact-fp32 MFMA GEMM (``mel_spectrogram``: the windowed DFT as a matrix, no vendor FFT; HIP tensors only).  The plain torch-ops form
(``torch.stft``) is test infrastructure (tests/torch_ops_mfcc.py) the HIP products are checked against.
    MelSpectrogram: torch.stft(center=True, pad_mode="reflect", hann window of win_length, onesided) -> |.|^2 ->
                    HTK mel filter bank (n_mels=128, f_min=0, f_max=sr/2, norm=None)
    AmplitudeToDB("power", top_db=80): 10 log10(max(x, 1e-10)), floored at (max over the whole (batch, mel, time) input) - 80
                    (torchaudio treats a 3-D input as one packed item; per item only for 4-D input)
    DCT-II, norm="ortho", first n_mfcc coefficients
``Wav2VecExtractor`` (extractor.py:62-96) is the wav2vec2 encoder of ``model/wav2vec.py`` (HF key names: load
``facebook/wav2vec2-base-960h`` with ``extractor.model.load_state_dict``; nothing is downloaded here) behind the resampler
(``data.resample_sinc_hann``, parity-unpinned) and the HF processor's normalisation; everything after the resampler is pinned
against the installed HF modules by fixture G12 (tests/golden/gen_golden.py).
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels) triangular filters on the HTK mel scale, no area normalisation (torchaudio F.melscale_fbanks)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def create_dct_ortho(n_mfcc: int, n_mels: int) -> torch.Tensor:
    """(n_mels, n_mfcc) DCT-II matrix with orthonormal scaling (torchaudio F.create_dct(norm="ortho"))."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


class MFCCExtractor(nn.Module):
    """Input (batch, time) -> output (batch, out_dim, n_mfcc); same constructor as extractor.py:16-35."""

    def __init__(self, sample_rate: int, n_feature: int, out_dim: int, win_length: int, hop_length: int = None, n_fft: int = None,
                 n_mels: int = 128, top_db: float = 80.0):
        super().__init__()
        self.sample_rate = sample_rate
        self.n_mfcc = n_feature
        self.out_dim = out_dim
        self.win_length = win_length
        self.hop_length = hop_length if hop_length else win_length // 2          # extractor.py:29-31
        self.n_fft = n_fft if n_fft else win_length                               # extractor.py:32
        self.top_db = top_db
        if self.n_mfcc > n_mels:
            raise ValueError("Cannot select more MFCC coefficients than # mel bins")
        self.register_buffer("window", torch.hann_window(self.win_length), persistent=False)
        self.register_buffer("fb", melscale_fbanks_htk(self.n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate), persistent=False)
        self.register_buffer("dct_mat", create_dct_ortho(self.n_mfcc, n_mels), persistent=False)
        self._first = True

    def _basis(self, device):
        """(2 * nfp, win_length) real DFT basis restricted to the support of the centred window: row f = w[k] cos(2 pi f (k + off) / n_fft),
        row nfp + f = -w[k] sin(...), f < n_fft/2 + 1 (rows up to nfp = the next multiple of 4 are zero), built in float64."""
        b = getattr(self, "_dft_basis", None)
        if b is None or b.device != device:
            nf = self.n_fft // 2 + 1
            nfp = (nf + 3) // 4 * 4
            off = (self.n_fft - self.win_length) // 2
            k = torch.arange(self.win_length, dtype=torch.float64) + off
            ang = 2.0 * math.pi * torch.arange(nf, dtype=torch.float64).view(-1, 1) * k.view(1, -1) / self.n_fft
            w = self.window.detach().double().cpu().view(1, -1)
            basis = torch.zeros(2 * nfp, self.win_length, dtype=torch.float64)
            basis[:nf] = w * torch.cos(ang)
            basis[nfp:nfp + nf] = -w * torch.sin(ang)
            fbp = torch.zeros(self.fb.shape[1], nfp)                               # (n_mels, nfp): mel filters as a Linear weight
            fbp[:, :nf] = self.fb.detach().cpu().t()
            b = self._dft_basis = basis.float().to(device)
            self._fb_lin = fbp.contiguous().to(device)
            self._dct_lin = self.dct_mat.detach().t().contiguous().to(device)      # (n_mfcc, n_mels)
            self._nfp = nfp
        return b

    def mel_spectrogram(self, x: torch.Tensor) -> torch.Tensor:
        """The mel power spectrogram with the STFT as a matrix product: frames of the reflect-padded signal (only the win_length
        samples under the centred window) against the windowed DFT basis, then |.|^2 and the mel filter bank - three launches of the
        exact-fp32 MFMA GEMM (a2f_gemm_f32) instead of the vendor FFT.  (batch, time) -> (batch, n_mels, frames)."""
        from . import ops
        be = ops.backend()
        basis = self._basis(x.device)
        B, N = x.shape
        pad, off, nfp = self.n_fft // 2, (self.n_fft - self.win_length) // 2, self._nfp
        xp = F.pad(x.float().unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
        frames = xp.unfold(1, self.n_fft, self.hop_length)[:, :, off:off + self.win_length].contiguous()       # (B, frames, win_length)
        nfr = frames.shape[1]
        M = B * nfr
        spec = torch.empty(M, 2 * nfp, device=x.device, dtype=torch.float32)
        ops.linear_fwd(be, frames.view(M, self.win_length), basis, None, spec, M, 2 * nfp, self.win_length)
        power = spec[:, :nfp] * spec[:, :nfp] + spec[:, nfp:] * spec[:, nfp:]                                  # columns >= n_freqs are zero
        mel = torch.empty(M, self._fb_lin.shape[0], device=x.device, dtype=torch.float32)
        ops.linear_fwd(be, power.contiguous(), self._fb_lin, None, mel, M, self._fb_lin.shape[0], nfp)
        return mel.view(B, nfr, -1).transpose(1, 2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """One path: the three matrix products run on the library's GEMM through ``ops.backend()`` (HIP tensors; off-device tensors fail in
        ``ops._chk`` like everywhere else in the package).  The torch.stft restatement the tests compare against lives in tests/torch_ops_mfcc.py."""
        mel = self.mel_spectrogram(x)
        db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
        # torchaudio.functional.amplitude_to_DB packs a (batch, mel, time) input as ONE (1, batch, mel, time) item: the top_db
        # floor is the maximum over the whole batch minus top_db (per item only for 4-D input).  extractor.py:47 feeds (batch, time)
        # audio, so the spectrogram is 3-D here.
        red = (-3, -2, -1) if db.dim() == 3 else (-2, -1)
        db = torch.maximum(db, db.amax(dim=red, keepdim=True) - self.top_db)
        from . import ops
        dbt = db.transpose(-1, -2).contiguous()                                    # (batch, frames, n_mels)
        mfcc = torch.empty(dbt.shape[0], dbt.shape[1], self.n_mfcc, device=x.device, dtype=torch.float32)
        ops.linear_fwd(ops.backend(), dbt.view(-1, dbt.shape[2]), self._dct_lin, None, mfcc.view(-1, self.n_mfcc), dbt.shape[0] * dbt.shape[1],
                       self.n_mfcc, dbt.shape[2])
        if self.out_dim != mfcc.shape[1]:                                          # extractor.py:49-58
            if self._first:
                logging.warning(f"MFCCExtractor: Got tensor shape {mfcc.shape}, resizing to {self.out_dim} using bilinear interpolation")
                self._first = False
            mfcc = F.interpolate(mfcc.unsqueeze(1), size=(self.out_dim, self.n_mfcc), mode="bilinear").squeeze(1)
        return mfcc


class Wav2VecExtractor(nn.Module):
    """Input (batch, time) at ``sample_rate`` -> (batch, out_dim, n_feature); same constructor as extractor.py:63-83.

    forward (extractor.py:84-96): resample to 16 kHz; the HF processor's zero-mean / unit-variance normalisation - computed over
    the WHOLE (batch, time) tensor, because the processor receives one tensor, wraps it as a single example and the reference
    then takes ``input_values[0]``; the plain wav2vec2 forward (49 Hz frames, no interpolation); transpose to (batch, 768, L);
    bilinear resize to (out_dim, n_feature) unless out_dim == 768.  The convolutional feature encoder is frozen
    (``freeze_feature_encoder``).  On HIP tensors the encoder runs the hand-written path of ``model/wav2vec.py``."""

    def __init__(self, sample_rate: int, n_feature: int, out_dim: int, *args, **kwargs):
        super().__init__()
        from .model.wav2vec import Wav2Vec2Model
        self.ori_sample_rate = sample_rate
        self.sample_rate = 16000
        self.out_dim = out_dim
        self.n_feature = n_feature
        self.model = Wav2Vec2Model()
        for p in self.model.feature_extractor.parameters():               # freeze_feature_encoder()
            p.requires_grad_(False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .data import resample_sinc_hann
        x = resample_sinc_hann(x, self.ori_sample_rate, self.sample_rate)
        x = x.to(torch.float32)
        x = (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + 1e-7)      # Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm on one example
        h = self.model(x, None).last_hidden_state                          # (batch, L, 768)
        h = h.transpose(1, 2)
        if self.out_dim != h.shape[1]:
            h = F.interpolate(h.unsqueeze(1), size=(self.out_dim, self.n_feature), mode="bilinear").squeeze(1)
        return h


def get_extractor(extractor: str):
    """lightning_model.py:61-67."""
    if extractor == "mfcc":
        return MFCCExtractor
    if extractor == "wav2vec":
        return Wav2VecExtractor
    raise KeyError(f"unknown feature extractor '{extractor}' (available: mfcc, wav2vec)")
