#!/usr/bin/env python3
"""Golden vectors for the data-side helpers (a2f_amd/data.py), generated from the reference's own
functions.  Runs only in the build container (needs /root/reference); the fixture it writes is data.

/root/reference/src/dataset/vocaset.py imports torchaudio, lightning and rich at module level, none of which is needed by
the pure-numpy functions pinned here, so empty stand-in modules are registered for the import (harness side only).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
for name in ("torchaudio", "torchaudio.functional", "lightning", "rich"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        sys.modules[name] = m
sys.modules["lightning"].LightningDataModule = object
sys.modules["rich"].print = print
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
from src.dataset import vocaset as ref          # noqa: E402

import a2f_amd                                   # noqa: E402,F401
from a2f_amd import data as mine                 # noqa: E402

rng = np.random.default_rng(20240607)
out = {}
cases = []
for ci, (n, dtype, sr, length) in enumerate([(30000, np.int16, 22000, 0.52), (5000, np.float32, 22000, 0.05), (16000, np.int16, 16000, 0.05),
                                              (400, np.int16, 22000, 0.05), (22000, np.float32, 22000, 0.025)]):
    audio = (rng.integers(-20000, 20000, n).astype(np.int16) if dtype == np.int16 else rng.standard_normal(n).astype(np.float32))
    out[f"c{ci}::audio"] = audio
    n_frames = n * 60 // sr
    for idx in sorted({0, 7, max(n_frames - 1, 0), n_frames + 3}) if ci else (0, 41, n_frames + 30):
        for shift in ((0, -200, 500, 137) if ci else (0, -500)):
            r = ref.get_audio_fragment(audio, idx, fps=60, sample_rate=sr, length=length, shift=shift)
            q = mine.get_audio_fragment(audio, idx, fps=60, sample_rate=sr, length=length, shift=shift)
            assert (r is None) == (q is None)
            if r is not None:
                assert r.dtype == q.dtype and np.array_equal(r, q)
                out[f"c{ci}::frag::{idx}::{shift}"] = r
            cases.append((ci, idx, shift, sr, length, r is None))
out["cases"] = np.array([(c[0], c[1], c[2], c[3], int(round(c[4] * 1000)), int(c[5])) for c in cases], dtype=np.int64)
pcm = rng.integers(-32768, 32767, 4096).astype(np.int16)
out["norm::in"] = pcm
out["norm::out"] = ref.normalize_audio(pcm)
assert np.array_equal(out["norm::out"], mine.normalize_audio(pcm))
ids = [*ref.training_subject, *ref.validation_subject, *ref.test_subject]
assert ids == mine.ALL_SUBJECTS
out["one_hot"] = np.stack([ref.get_human_id_one_hot(h) for h in ids])
assert np.array_equal(out["one_hot"], np.stack([mine.human_id_one_hot(h) for h in ids]))
np.savez_compressed(os.path.join(HERE, "g9_vocaset_helpers.npz"), **out)
print("wrote g9_vocaset_helpers.npz:", len(cases), "fragment cases,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")

# ---------------------------------------------------------------------------------------------------------------------------
# G11: the loader classes (DataSplitRecorder, ClipVocaSet per-frame path) of the reference on a synthetic VOCASET directory
# ---------------------------------------------------------------------------------------------------------------------------
import tempfile                                    # noqa: E402
import contextlib, io                              # noqa: E402,E401
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vocaset_synth                               # noqa: E402
from a2f_amd import vocaset as mine_vs             # noqa: E402

out = {}
with tempfile.TemporaryDirectory() as d_ref, tempfile.TemporaryDirectory() as d_mine:
    vocaset_synth.build(d_ref)
    vocaset_synth.build(d_mine)
    for phase in ("train", "val", "test", "all"):
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref.ClipVocaSet(d_ref, phase=phase, split_frame=True)
        q = mine_vs.ClipVocaSet(d_mine, phase=phase, split_frame=True)
        assert [tuple(x) for x in r.datalist] == [tuple(x) for x in q.datalist], phase
        out[f"{phase}::human_id"] = np.array([x[0] for x in r.datalist])
        out[f"{phase}::sentence_id"] = np.array([x[1] for x in r.datalist])
        out[f"{phase}::index"] = np.array([[x[2], x[3]] for x in r.datalist], dtype=np.int64)
        for i in sorted({0, len(r) // 2, len(r) - 1}):
            a, b = r[i], q[i]
            for k in ("audio", "verts", "template_vert", "one_hot"):
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (phase, i, k)
                out[f"{phase}::item{i}::{k}"] = a[k]
    with contextlib.redirect_stdout(io.StringIO()):
        r = ref.ClipVocaSet(d_ref, phase="test", split_frame=True)
    q = mine_vs.ClipVocaSet(d_mine, phase="test", split_frame=True)
    fr, fq = r.get_framedatas("FaceTalk_170809_00138_TA", "sentence21"), q.get_framedatas("FaceTalk_170809_00138_TA", "sentence21")
    assert len(fr) == len(fq) and all(np.array_equal(x["audio"], y["audio"]) for x, y in zip(fr, fq))
    out["framedatas::count"] = np.int64(len(fr))
    out["framedatas::audio0"] = fr[0]["audio"]
    # whole clips (split_frame=False, the FaceFormer branch: vocaset.py:271-289).  The 22 kHz -> 16 kHz step is torchaudio's (absent here:
    # PARITY UNPINNED, a2f_amd.data.resample_sinc_hann restates it) - both sides get the SAME stand-in resampler, so that everything around
    # it is pinned: the unique (subject, sentence) list, int16 normalisation BEFORE the resampler, the call's (22000, 16000) arguments, the
    # float32 cast, and the stack of every mesh of the sentence in wav_seq_to_idx order.
    import torch                                   # noqa: E402

    def stand_in_resample(wav, orig, new):
        assert (orig, new) == (22000, 16000) and wav.dtype == torch.float32
        n = -(-wav.shape[-1] * new // orig)
        return wav[..., (torch.arange(n) * orig) // new] * 0.5 + 0.25

    sys.modules["torchaudio"].functional = sys.modules["torchaudio.functional"]
    sys.modules["torchaudio.functional"].resample = stand_in_resample
    ref.torchaudio = sys.modules["torchaudio"]
    mine_vs.resample_sinc_hann = stand_in_resample
    for phase in ("train", "test"):
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref.ClipVocaSet(d_ref, phase=phase, split_frame=False)
        q = mine_vs.ClipVocaSet(d_mine, phase=phase, split_frame=False)
        keys = sorted(tuple(x) for x in r.datalist)
        assert keys == sorted(tuple(x) for x in q.datalist) and len(set(keys)) == len(keys), phase      # the reference keeps set order: compared as sets
        out[f"clips::{phase}::keys"] = np.array(["|".join(k) for k in keys])
        for key in (keys[0], keys[len(keys) // 2], keys[-1]):
            a, b = r[[tuple(x) for x in r.datalist].index(key)], q[[tuple(x) for x in q.datalist].index(key)]
            for k in ("audio", "verts", "template_vert", "one_hot"):
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (phase, key, k)
                out[f"clips::{phase}::{'|'.join(key)}::{k}"] = a[k]
        fr, fq = r.get_framedatas(*keys[0]), q.get_framedatas(*keys[0])
        assert len(fr) == len(fq) == 1 and np.array_equal(fr[0]["audio"], fq[0]["audio"]) and np.array_equal(fr[0]["verts"], fq[0]["verts"])
np.savez_compressed(os.path.join(HERE, "g11_vocaset_loader.npz"), **out)
print("wrote g11_vocaset_loader.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")
