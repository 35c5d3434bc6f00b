"""TEST INFRASTRUCTURE: the torch-ops restatement of a2f_amd.extractor.MFCCExtractor (torch.stft form of torchaudio's published MFCC defaults,
/root/reference/src/model/extractor.py:10-59).  The product evaluates the same definition as three matrix products on its own GEMM; this file is
what those are compared with (CPU: through the float64 emulator; GPU: HIP products vs torch.stft on the device and on the CPU)."""
import torch
import torch.nn.functional as F


def mel_spectrogram_stft(ex, x):
    spec = torch.stft(x, ex.n_fft, hop_length=ex.hop_length, win_length=ex.win_length, window=ex.window.to(x.device), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real * spec.real + spec.imag * spec.imag                     # (batch, n_freqs, frames)
    return torch.matmul(power.transpose(-1, -2), ex.fb.to(x.device)).transpose(-1, -2)   # (batch, n_mels, frames)


def mfcc_torch_ops(ex, x):
    mel = mel_spectrogram_stft(ex, x)
    db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    red = (-3, -2, -1) if db.dim() == 3 else (-2, -1)
    db = torch.maximum(db, db.amax(dim=red, keepdim=True) - ex.top_db)
    mfcc = torch.matmul(db.transpose(-1, -2), ex.dct_mat.to(x.device))
    if ex.out_dim != mfcc.shape[1]:
        mfcc = F.interpolate(mfcc.unsqueeze(1), size=(ex.out_dim, ex.n_mfcc), mode="bilinear").squeeze(1)
    return mfcc
