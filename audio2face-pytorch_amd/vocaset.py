"""VOCASET loader either side of the hot path, Lightning / torchaudio free.

Restated from /root/reference/src/dataset/vocaset.py: the split builder ``DataSplitRecorder`` (:80-182: every
(subject, sentence, clip index, vertex row) of ``subj_seq_to_idx.pkl`` goes to train / val / test by the subject and sentence
lists of :31-47 and is written as three CSV files under ``<datapath>/split``), the dataset ``ClipVocaSet`` (:185-323: one
0.52 s audio window + one mesh per item when ``split_frame``, else one whole clip resampled 22 kHz -> 16 kHz with all its
meshes), and the data module ``VocaDataModule`` (:326-398: train loader shuffled with drop_last, val / test loaders in order,
a predict loader over one sentence).  Files it reads (same names as the reference): ``templates.pkl``,
``raw_audio_fixed.pkl``, ``data_verts.npy`` (memory-mapped), ``subj_seq_to_idx.pkl``.

The per-frame path and the split lists are pinned against the reference's own classes on a synthetic dataset directory
(tests/golden/gen_golden_data.py -> g11_vocaset_loader.npz).  The whole-clip path uses ``data.resample_sinc_hann`` where the
reference calls ``torchaudio.functional.resample`` (torchaudio is absent here: that one step is parity-unpinned, see data.py).
"""
from __future__ import annotations

import os
import pickle
import random
from typing import Dict, List, Literal, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .data import (TEST_SUBJECTS, TRAINING_SUBJECTS, VALIDATION_SUBJECTS, get_audio_fragment, human_id_one_hot, normalize_audio,
                   resample_sinc_hann)

TRAINING_SENTENCES = [f"sentence{i:02d}" for i in range(1, 41)]            # vocaset.py:41
VALIDATION_SENTENCES = [f"sentence{i:02d}" for i in range(21, 41)]         # vocaset.py:46
SPLIT_COLUMNS = ["human_id", "sentence_id", "clip_index", "data_verts_index"]


def load_pickle(path: str):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")                          # vocaset.py:16-19 (the dataset pickles are python-2 era)


class DataSplitRecorder:
    """vocaset.py:80-182."""

    def __init__(self, write: bool = True) -> None:
        self.write = write
        self.train_list: List = []
        self.val_list: List = []
        self.test_list: List = []

    def add(self, human_id: str, sentence_id: str, clip_index: int, data_verts_index: int):
        if not self.write:
            raise ValueError("This instance is read only")
        rec = (human_id, sentence_id, clip_index, data_verts_index)
        if human_id in TRAINING_SUBJECTS and sentence_id in TRAINING_SENTENCES:
            self.train_list.append(rec)
        elif human_id in VALIDATION_SUBJECTS and sentence_id in VALIDATION_SENTENCES:
            self.val_list.append(rec)
        else:                                                             # everything else, including val subjects' sentences 1-20
            self.test_list.append(rec)

    def save(self, path: str):
        if not self.write:
            raise ValueError("This instance is read only")
        import pandas as pd
        for ls, name in ((self.train_list, "train_list.csv"), (self.val_list, "val_list.csv"), (self.test_list, "test_list.csv")):
            pd.DataFrame(ls, columns=SPLIT_COLUMNS).to_csv(os.path.join(path, name), index=False)

    @staticmethod
    def build(raw_audio, subj_seq_to_idx, *, save_path: str):
        save_path = os.path.join(save_path, "split")
        os.makedirs(save_path, exist_ok=False)
        rec = DataSplitRecorder()
        for clip_name, clip_data in raw_audio.items():
            if clip_name not in subj_seq_to_idx:
                continue
            for sentence_id in clip_data:
                if sentence_id not in subj_seq_to_idx[clip_name]:
                    continue
                for clip_index, seq_num in subj_seq_to_idx[clip_name][sentence_id].items():
                    rec.add(clip_name, sentence_id, clip_index, seq_num)
        rec.save(save_path)

    @staticmethod
    def exists(path: str) -> bool:
        path = os.path.join(path, "split")
        return all(os.path.exists(os.path.join(path, n)) for n in ("train_list.csv", "val_list.csv", "test_list.csv"))

    @classmethod
    def load(cls, path: str):
        import pandas as pd
        path = os.path.join(path, "split")
        rec = cls(write=False)
        rec.train_list, rec.val_list, rec.test_list = (pd.read_csv(os.path.join(path, n), header=0).to_numpy().tolist()
                                                       for n in ("train_list.csv", "val_list.csv", "test_list.csv"))
        return rec

    def get_list(self, phase: str = "all"):
        return {"train": self.train_list, "val": self.val_list, "test": self.test_list}.get(phase, self.train_list + self.val_list + self.test_list)


class ClipVocaSet(Dataset):
    """vocaset.py:185-323.  Items are dictionaries ``audio`` / ``verts`` / ``template_vert`` / ``one_hot`` of numpy arrays (the
    default collate of ``DataLoader`` stacks them into the batch dictionaries ``TrainStep.unpack_batch`` takes)."""

    def __init__(self, datapath: str, phase: Literal["train", "val", "test", "all"] = "all", random_shift: bool = False,
                 split_frame: bool = True, normalize_audio: bool = True):
        if not split_frame:
            assert random_shift is False, "random_shift is not supported when split_frame is False"
        self.phase, self.random_shift, self.split_frame, self.normalize_audio = phase, random_shift, split_frame, normalize_audio
        self.datapath = os.path.abspath(datapath)
        self.template_verts = load_pickle(os.path.join(self.datapath, "templates.pkl"))
        self.raw_audio = load_pickle(os.path.join(self.datapath, "raw_audio_fixed.pkl"))
        self.data_verts = np.load(os.path.join(self.datapath, "data_verts.npy"), mmap_mode="r")
        self.wav_seq_to_idx = load_pickle(os.path.join(self.datapath, "subj_seq_to_idx.pkl"))
        if not DataSplitRecorder.exists(self.datapath):
            DataSplitRecorder.build(self.raw_audio, self.wav_seq_to_idx, save_path=self.datapath)
        self.split_recorder = DataSplitRecorder.load(self.datapath)
        self.datalist_raw = self.split_recorder.get_list(phase)
        if split_frame:
            self.datalist = self.datalist_raw
        else:                                                             # one item per (subject, sentence); the reference keeps set order
            self.datalist = list({(d[0], d[1]) for d in self.datalist_raw})

    def __repr__(self) -> str:
        return f"ClipVocaSet(\n    phase={self.phase},\n    datapath={self.datapath},\n    len={len(self.datalist)}\n)"

    def get_single_item(self, key) -> Dict[str, np.ndarray]:
        human_id, sentence_id, audio_index, data_verts_index = key
        audio = self.raw_audio[human_id][sentence_id]["audio"]
        sr = self.raw_audio[human_id][sentence_id]["sample_rate"]
        verts = self.data_verts[data_verts_index]
        shift = random.randint(-500, 500) if self.random_shift and self.phase == "train" else 0        # vocaset.py:254-257
        clip = get_audio_fragment(audio, audio_index, fps=60, sample_rate=sr, length=0.52, shift=shift)
        if self.normalize_audio:
            clip = normalize_audio(clip)
        return {"audio": clip, "verts": verts.astype(np.float32), "template_vert": self.template_verts[human_id].astype(np.float32),
                "one_hot": human_id_one_hot(human_id).astype(np.float32)}

    def get_whole_clip(self, key) -> Dict[str, np.ndarray]:
        human_id, sentence_id = key
        clip = self.raw_audio[human_id][sentence_id]["audio"]
        verts = [self.data_verts[seq_num] for seq_num in self.wav_seq_to_idx[human_id][sentence_id].values()]
        if self.normalize_audio:
            clip = normalize_audio(clip)
        clip = resample_sinc_hann(torch.from_numpy(np.asarray(clip, dtype=np.float32)), 22000, 16000).numpy().astype(np.float32)   # :276-281
        return {"audio": clip, "verts": np.stack(verts).astype(np.float32), "template_vert": self.template_verts[human_id].astype(np.float32),
                "one_hot": human_id_one_hot(human_id).astype(np.float32)}

    def __len__(self):
        return len(self.datalist)

    def __getitem__(self, idx):
        return self.get_single_item(self.datalist[idx]) if self.split_frame else self.get_whole_clip(self.datalist[idx])

    def get_framedatas(self, target_human_id: str, target_sentence_id: str):
        """vocaset.py:300-323 (the reference's ``sorted(...)`` result is discarded there, so list order = split-file order)."""
        if self.split_frame:
            return [self.get_single_item(d) for d in self.datalist if d[0] == target_human_id and d[1] == target_sentence_id]
        for d in self.datalist:
            if d[0] == target_human_id and d[1] == target_sentence_id:
                return [self.get_whole_clip(d)]
        return []


class VocaDataModule:
    """vocaset.py:326-398 without the Lightning base class: ``setup()`` then the four loader factories."""

    def __init__(self, datapath: str, batch_size: int = 32, num_workers: int = 4, random_shift: bool = False, split_frame: bool = True):
        self.datapath, self.batch_size, self.num_workers = datapath, batch_size, num_workers
        self.random_shift, self.split_frame = random_shift, split_frame
        self.train_dataset: Optional[ClipVocaSet] = None

    def setup(self, stage: Optional[str] = None):
        mk = lambda phase: ClipVocaSet(self.datapath, phase=phase, random_shift=self.random_shift, split_frame=self.split_frame)
        self.train_dataset, self.val_dataset, self.test_dataset = mk("train"), mk("val"), mk("test")

    def _loader(self, ds, shuffle: bool, drop_last: bool):
        return DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle, num_workers=self.num_workers, pin_memory=torch.cuda.is_available(),
                          drop_last=drop_last)

    def train_dataloader(self):
        return self._loader(self.train_dataset, True, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False, False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False, False)

    def predict_dataloader(self, human_id: str, sentence_id: str):
        return DataLoader(self.test_dataset.get_framedatas(human_id, sentence_id), batch_size=self.batch_size, shuffle=False,
                          num_workers=self.num_workers)
