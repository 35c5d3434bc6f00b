"""CPU: the Lightning-free fit loop (a2f_amd.fit; reference: train.py:28-67, lightning_model.py:127-207) - callbacks, checkpoint
layout and round trip, the FaceFormer overrides, and the 3-step training trajectory generated from the reference (G10),
with the C ABI emulated (tests/emu_backend.py)."""
import os

import pytest
import torch

import a2f_amd
from a2f_amd import fit as F
from a2f_amd import ops
from a2f_amd.registry import ExpConfig
from a2f_amd.trainer import TrainStep
from emu_backend import EmuBackend
from golden_util import load, packed_error
from oracle import weights as W

NV = 15069


@pytest.fixture()
def emu():
    old = ops.set_backend(EmuBackend())
    yield
    ops.set_backend(old)


def _cfg(**kw):
    base = dict(batch_size=4, modelname="voca", vertex_count=NV, one_hot_size=12, split_frame=True, percision="32", lr=1e-3,
                feature_extractor=None, sample_rate=22000, n_feature=16, out_dim=29, win_length=440)
    base.update(kw)
    return ExpConfig(**base)


def test_early_stopping_and_checkpoint_callbacks(tmp_path):
    es = F.EarlyStopping(patience=3)
    seq = [5.0, 4.0, 4.5, 4.2, 4.0, 3.9]            # improves at epochs 0 and 1; 4.0 at epoch 4 is NOT < 4.0: third epoch without improvement
    stopped_at = None
    for e, v in enumerate(seq):
        if es.update(v, e):
            stopped_at = e
            break
    assert stopped_at == 4 and es.stopped_epoch == 4 and es.best == 4.0
    es2 = F.EarlyStopping(patience=2)
    assert [es2.update(v, e) for e, v in enumerate([3.0, 2.0, 2.5, 1.0, 1.5])] == [False] * 5     # an improvement resets the counter
    mc = F.ModelCheckpoint(str(tmp_path / "checkpoints"))
    saved = []
    for e, v in enumerate([3.0, 2.0, 2.5, 1.0]):
        mc.update(v, e, 10 * (e + 1), lambda p: (open(p, "w").write("x"), saved.append(p)))
    assert [os.path.basename(p) for p in saved] == ["epoch=0-step=10.ckpt", "epoch=1-step=20.ckpt", "epoch=3-step=40.ckpt"]
    assert os.listdir(tmp_path / "checkpoints") == ["epoch=3-step=40.ckpt"]                        # save_top_k = 1, save_last = False
    assert F.best_checkpoint(str(tmp_path / "checkpoints")).endswith("epoch=3-step=40.ckpt")


def test_faceformer_overrides():
    cfg = F.apply_model_overrides(_cfg(modelname="faceformer", batch_size=128, feature_extractor="mfcc"))
    assert cfg.batch_size == 1 and cfg.split_frame is False and cfg.feature_extractor is None              # train.py:30-34
    cfg = F.apply_model_overrides(_cfg(batch_size=128, feature_extractor="mfcc"))
    assert cfg.batch_size == 128 and cfg.feature_extractor == "mfcc"


def _g10_batches(g):
    B = int(g["B"])
    for it in range(3):
        tv = W.formula_input(f"g10.template{it}", (B, 5023, 3), 0.05)
        yield {"audio": W.formula_input(f"g10.x{it}", (B, 29, 16)), "one_hot": torch.from_numpy(g["one_hot"]),
               "template_vert": tv, "verts": tv + W.formula_input(f"g10.gt{it}", (B, 5023, 3), 0.005)}


def test_training_trajectory_matches_reference_g10(emu):
    """Three TrainStep steps (x100 unpack, fused forward/loss/backward, flat Adam with weight decay lr/10) reproduce the
    loss / err trajectory and the final parameters of the reference model + loss + torch.optim.Adam (fixture G10)."""
    g = load("g10_voca_train_trajectory")
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    m.load_state_dict(W.formula_state_dict(W.voca_shapes(NV), gains={"decoder.4": 4.0}))
    step = TrainStep(m, a2f_amd.VocaLoss(), lr=float(g["lr"]))
    for it, batch in enumerate(_g10_batches(g)):
        out = step(*TrainStep.unpack_batch(batch))
        for k in ("loss", "rec_loss", "vel_loss", "err"):
            ref = float(g[f"step{it}::{k}"])
            assert abs(float(out[k]) - ref) <= 2e-5 * abs(ref), (it, k, float(out[k]), ref)
    for k, p in m.named_parameters():
        assert packed_error(g, "final::" + k, p) < 1e-5, k


def test_checkpoint_round_trip_and_layout(emu, tmp_path):
    g = load("g10_voca_train_trajectory")
    cfg = _cfg()
    m = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    step = TrainStep(m, a2f_amd.VocaLoss(), lr=cfg.lr)
    batches = list(_g10_batches(g))
    step(*TrainStep.unpack_batch(dict(batches[0])))
    path = str(tmp_path / "epoch=0-step=1.ckpt")
    F.save_checkpoint(path, m, step, cfg, 0, 1)
    ck = torch.load(path, weights_only=False)
    assert set(ck) >= {"state_dict", "hyper_parameters", "epoch", "global_step", "pytorch-lightning_version", "optimizer_states"}
    assert all(k.startswith("model.") for k in ck["state_dict"]) and ck["hyper_parameters"]["config"].modelname == "voca"
    assert set(k[len("model."):] for k in ck["state_dict"]) == set(W.voca_shapes(NV))                      # the reference's key names
    # resume in a fresh model/optimizer: the second step must be bit-identical to continuing the first run
    out_a = step(*TrainStep.unpack_batch(dict(batches[1])))
    m2 = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    step2 = TrainStep(m2, a2f_amd.VocaLoss(), lr=cfg.lr)
    F.load_checkpoint(path, m2, step2)
    out_b = step2(*TrainStep.unpack_batch(dict(batches[1])))
    assert float(out_a["loss"]) == float(out_b["loss"]) and step2.opt.step_count == 2
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), k


def test_fit_loop_validates_checkpoints_and_reloads_best(emu, tmp_path):
    """fit(): epochs of TrainStep + validation (eval mode, val/err), best-only checkpoint, early stop, best reloaded at the end."""
    cfg = _cfg(lr=1e-3)
    gen = torch.Generator().manual_seed(5)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        tv = torch.randn(4, 5023, 3, generator=g) * 0.05
        oh = torch.zeros(4, 12); oh[torch.arange(4), torch.arange(4) % 8] = 1
        return {"audio": torch.randn(4, 29, 16, generator=g), "one_hot": oh, "template_vert": tv, "verts": tv + torch.randn(4, 5023, 3, generator=g) * 0.005}
    logs = []
    model, hist = F.fit(cfg, lambda e: (batch(100 + 2 * e + i) for i in range(2)), lambda e: (batch(7),), "cpu", str(tmp_path), max_epochs=3,
                        patience=5, log=logs.append)
    assert len(hist) == 3 and all("val/err" in h and "train/err" in h for h in hist)
    best = min(range(3), key=lambda i: hist[i]["val/err"])
    assert F.fit.last_run_dir == str(tmp_path / "version_0")                                              # TensorBoardLogger's layout: one fresh version_N per run
    files = os.listdir(tmp_path / "version_0" / "checkpoints")
    assert files == [f"epoch={best}-step={2 * (best + 1)}.ckpt"]                                          # written atomically: no .tmp left behind
    ck = torch.load(tmp_path / "version_0" / "checkpoints" / files[0], weights_only=False)
    for k, p in model.state_dict().items():
        assert torch.equal(p.cpu(), ck["state_dict"]["model." + k]), k                                      # best checkpoint is what fit() returns


def test_fit_second_run_never_reloads_a_stale_checkpoint(emu, tmp_path):
    """A leftover checkpoint with a larger step= from an earlier run in the same experiment directory must not be what fit() reloads."""
    cfg = _cfg(lr=1e-3)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        tv = torch.randn(4, 5023, 3, generator=g) * 0.05
        oh = torch.zeros(4, 12); oh[torch.arange(4), torch.arange(4) % 8] = 1
        return {"audio": torch.randn(4, 29, 16, generator=g), "one_hot": oh, "template_vert": tv, "verts": tv + torch.randn(4, 5023, 3, generator=g) * 0.005}
    os.makedirs(tmp_path / "version_0" / "checkpoints")
    (tmp_path / "version_0" / "checkpoints" / "epoch=9-step=999999.ckpt").write_bytes(b"not a checkpoint")
    model, hist = F.fit(cfg, lambda e: (batch(100 + e),), lambda e: (batch(7),), "cpu", str(tmp_path), max_epochs=1, log=lambda *_: None)
    assert F.fit.last_run_dir == str(tmp_path / "version_1") and len(os.listdir(tmp_path / "version_1" / "checkpoints")) == 1


def _fit_rank(rank, world, port, tmp, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_backend import EmuBackend
    ops.set_backend(EmuBackend())
    cfg = _cfg(lr=1e-3)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        tv = torch.randn(4, 5023, 3, generator=g) * 0.05
        oh = torch.zeros(4, 12); oh[torch.arange(4), torch.arange(4) % 8] = 1
        return {"audio": torch.randn(4, 29, 16, generator=g), "one_hot": oh, "template_vert": tv, "verts": tv + torch.randn(4, 5023, 3, generator=g) * 0.005}
    torch.manual_seed(0)
    # different validation data per rank: without the all-reduce of val/err the ranks would disagree on which epoch is best
    model, hist = F.fit(cfg, lambda e: (batch(100 + 10 * e + rank),), lambda e: (batch(7 + 50 * rank + e),), "cpu", tmp, max_epochs=3, log=lambda *_: None)
    q.put((rank, [h["val/err"] for h in hist], [h["saved"] for h in hist], F.fit.last_run_dir,
           float(sum(p.double().sum() for p in model.parameters()))))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_under_two_ranks_writes_once_and_agrees(tmp_path):
    """torchrun-style launch (gloo, world 2): every rank runs fit(); val/err is rank-averaged so decisions agree, rank 0 alone writes
    (one checkpoint file, no temp files), all ranks reload the same best checkpoint after the barrier."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fit_rank, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, v0, s0, d0, w0), (_, v1, s1, d1, w1) = res
    assert v0 == v1 and s0 == s1 and d0 == d1 and w0 == w1
    files = os.listdir(os.path.join(d0, "checkpoints"))
    assert len(files) == 1 and files[0].endswith(".ckpt")
