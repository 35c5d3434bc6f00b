"""CPU: the VOCASET loader (a2f_amd/vocaset.py; reference: src/dataset/vocaset.py:80-398) against fixture G11, which
tests/golden/gen_golden_data.py produced by running the REFERENCE's DataSplitRecorder / ClipVocaSet on the same synthetic dataset
directory (tests/vocaset_synth.py)."""
import os

import numpy as np
import torch

import a2f_amd  # noqa: F401
from a2f_amd import vocaset as V
from golden_util import load
import vocaset_synth


def test_split_lists_and_items_match_reference_g11(tmp_path):
    g = load("g11_vocaset_loader")
    d = vocaset_synth.build(str(tmp_path / "voca"))
    for phase in ("train", "val", "test", "all"):
        ds = V.ClipVocaSet(d, phase=phase, split_frame=True)
        assert [x[0] for x in ds.datalist] == list(g[f"{phase}::human_id"]) and [x[1] for x in ds.datalist] == list(g[f"{phase}::sentence_id"])
        assert np.array_equal(np.array([[x[2], x[3]] for x in ds.datalist], dtype=np.int64), g[f"{phase}::index"])
        for i in sorted({0, len(ds) // 2, len(ds) - 1}):
            item = ds[i]
            for k in ("audio", "verts", "template_vert", "one_hot"):
                assert item[k].dtype == g[f"{phase}::item{i}::{k}"].dtype and np.array_equal(item[k], g[f"{phase}::item{i}::{k}"]), (phase, i, k)
    assert sorted(os.listdir(os.path.join(d, "split"))) == ["test_list.csv", "train_list.csv", "val_list.csv"]
    test = V.ClipVocaSet(d, phase="test", split_frame=True)
    fr = test.get_framedatas("FaceTalk_170809_00138_TA", "sentence21")
    assert len(fr) == int(g["framedatas::count"]) and np.array_equal(fr[0]["audio"], g["framedatas::audio0"])
    # split rules (vocaset.py:99-110): a validation subject's sentences 1-20 land in the TEST list
    assert all(not (h in V.VALIDATION_SUBJECTS and s == "sentence01") for h, s, *_ in V.ClipVocaSet(d, phase="val").datalist)
    assert any(h in V.VALIDATION_SUBJECTS and s == "sentence01" for h, s, *_ in test.datalist)


def test_data_module_batches_and_whole_clips(tmp_path):
    d = vocaset_synth.build(str(tmp_path / "voca"))
    dm = V.VocaDataModule(d, batch_size=4, num_workers=0)
    dm.setup()
    batch = next(iter(dm.train_dataloader()))
    assert batch["audio"].shape == (4, 11440) and batch["audio"].dtype == torch.float32          # 0.52 s at 22 kHz (vocaset.py:253-260)
    assert batch["verts"].shape == (4, 17, 3) and batch["template_vert"].shape == (4, 17, 3) and batch["one_hot"].shape == (4, 12)
    assert len(dm.train_dataloader()) == len(dm.train_dataset) // 4                                 # drop_last
    assert sum(b["audio"].shape[0] for b in dm.val_dataloader()) == len(dm.val_dataset)             # keeps the ragged tail
    pred = list(dm.predict_dataloader("FaceTalk_170809_00138_TA", "sentence21"))
    assert sum(b["audio"].shape[0] for b in pred) == len(dm.test_dataset.get_framedatas("FaceTalk_170809_00138_TA", "sentence21"))
    # whole clips (FaceFormer, train.py:30-34): 22 kHz -> 16 kHz, every mesh of the sentence, batch 1
    clips = V.ClipVocaSet(d, phase="train", split_frame=False)
    assert len(clips) == len({(x[0], x[1]) for x in V.ClipVocaSet(d, phase="train").datalist})
    item = clips[0]
    h, s = clips.datalist[0]
    n22 = len(clips.raw_audio[h][s]["audio"])
    assert item["audio"].dtype == np.float32 and item["audio"].shape == (-(-16000 * n22 // 22000),)
    assert item["verts"].shape == (len(clips.wav_seq_to_idx[h][s]), 17, 3)
    try:
        V.ClipVocaSet(d, phase="train", split_frame=False, random_shift=True)
        assert False
    except AssertionError:
        pass
