"""A tiny synthetic VOCASET directory (same four files and layouts as the real dataset, vocaset.py:206-222) built from a seed:
used by tests/golden/gen_golden_data.py (which runs the REFERENCE loader on it) and by the loader tests (which run ours)."""
import os
import pickle

import numpy as np

SUBJECTS = ["FaceTalk_170728_03272_TA", "FaceTalk_170904_00128_TA", "FaceTalk_170811_03275_TA", "FaceTalk_170908_03277_TA",
            "FaceTalk_170809_00138_TA"]                                   # two training, two validation, one test subject
SENTENCES = ["sentence01", "sentence20", "sentence21", "sentence40"]


def build(path: str, seed: int = 2024, n_verts: int = 17):
    rng = np.random.default_rng(seed)
    os.makedirs(path, exist_ok=True)
    templates, raw_audio, seq_to_idx, verts = {}, {}, {}, []
    row = 0
    for s in SUBJECTS:
        templates[s] = rng.standard_normal((n_verts, 3)).astype(np.float64)
        raw_audio[s], seq_to_idx[s] = {}, {}
        for sent in SENTENCES:
            if s == SUBJECTS[1] and sent == "sentence40":
                continue                                                  # a sentence without audio
            n_frames = int(rng.integers(5, 12))
            n = n_frames * 22000 // 60 + int(rng.integers(0, 300))
            raw_audio[s][sent] = {"audio": rng.integers(-20000, 20000, n).astype(np.int16), "sample_rate": 22000}
            if s == SUBJECTS[0] and sent == "sentence20":
                continue                                                  # audio without meshes: must be skipped by the split builder
            seq_to_idx[s][sent] = {}
            for f in range(n_frames):
                seq_to_idx[s][sent][f] = row
                verts.append(rng.standard_normal((n_verts, 3)))
                row += 1
    pickle.dump(templates, open(os.path.join(path, "templates.pkl"), "wb"))
    pickle.dump(raw_audio, open(os.path.join(path, "raw_audio_fixed.pkl"), "wb"))
    pickle.dump(seq_to_idx, open(os.path.join(path, "subj_seq_to_idx.pkl"), "wb"))
    np.save(os.path.join(path, "data_verts.npy"), np.stack(verts))
    return path
