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
