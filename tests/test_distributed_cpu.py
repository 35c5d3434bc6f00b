"""CPU, world_size 2 (gloo): the data-parallel step = one all-reduce of the flat gradient buffer + fused Adam with the
1/world averaging folded in.  Two ranks on disjoint even-sized shards must land on the same parameters as one process on
the concatenated batch (VocaLoss means over equal shards average exactly; velocity pairs never straddle a shard).
The C ABI is emulated on CPU (tests/emu_backend.py); the HIP kernels themselves are covered by the -m gpu tests."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NV = 33


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(B):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 29, 16, generator=g)
    oh = torch.zeros(B, 12); oh[torch.arange(B), torch.arange(B) % 8] = 1
    t = torch.randn(B, NV // 3, 3, generator=g)
    gt = t + 0.3 * torch.randn(B, NV // 3, 3, generator=g)
    return x, oh, gt, t


def _run(rank, world, port, out_path, fused, overlap=None):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import a2f_amd
    from a2f_amd import ops
    from a2f_amd.trainer import TrainStep
    from emu_backend import EmuBackend
    if world > 1:
        torch.set_num_threads(1)
    ops.set_backend(EmuBackend())
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(11)
    model = a2f_amd.Voca(n_verts=NV, n_onehot=12)
    step = TrainStep(model, a2f_amd.VocaLoss(), lr=1e-3, fused=fused, overlap=overlap)
    assert overlap is None or step.overlap == (overlap and fused)
    x, oh, gt, t = _make_batch(8)
    per = 8 // world
    sl = slice(rank * per, (rank + 1) * per)
    for _ in range(2):
        out = step(x[sl], oh[sl], gt[sl], t[sl])
    if rank == 0:
        torch.save({"flat": step.opt.flat_p.clone(), "loss": float(out["loss"].detach())}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("fused,overlap", [(False, None), (True, False), (True, True)])
def test_dp2_gloo_matches_single_process(fused, overlap):
    """fused=True exercises the trainer's autograd-free direct step (grads written straight into the flat buffer);
    overlap=True its two-phase form: the head's gradient segment is all-reduced asynchronously while the trunk backward
    runs, the trunk segment afterwards (the default when distributed)."""
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "w1.pt"), os.path.join(d, "w2.pt")
        _run(0, 1, 0, p1, False)                 # reference run: plain autograd graph, one process, full batch
        from a2f_amd import ops
        ops.set_backend(None)
        mp.spawn(_run, args=(2, _free_port(), p2, fused, overlap), nprocs=2, join=True)
        a, b = torch.load(p1), torch.load(p2)
        rel = float((a["flat"] - b["flat"]).norm() / a["flat"].norm())
        assert rel < 1e-5, rel


def test_flat_adam_aliases_parameters_and_matches_torch_adam():
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import a2f_amd  # noqa: F401
    from a2f_amd import ops
    from a2f_amd.trainer import FlatAdam
    from emu_backend import EmuBackend
    old = ops.set_backend(EmuBackend())
    try:
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        ref.load_state_dict(m.state_dict())
        opt = FlatAdam(m.parameters(), lr=1e-2, weight_decay=1e-3)
        ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-3)
        x = torch.randn(4, 5)
        for _ in range(3):
            opt.zero_grad(); ropt.zero_grad()
            m(x).pow(2).sum().backward(); ref(x).pow(2).sum().backward()
            opt.step(); ropt.step()
        for a, b in zip(m.parameters(), ref.parameters()):
            assert a.data_ptr() >= opt.flat_p.data_ptr() and torch.allclose(a, b, atol=1e-6)
    finally:
        ops.set_backend(old)


# ---------------------------------------------------------------------------------------------------------------------------
# FaceFormer: sequences shard over ranks (the reference trains batch = 1 per sequence; S sequences per step here)
# ---------------------------------------------------------------------------------------------------------------------------
def _run_ff(rank, world, port, out_path, layers=0, skip_by_rank=False, S=4):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import types
    import a2f_amd
    from a2f_amd import ops
    from a2f_amd.trainer import TrainStep
    from emu_backend import EmuBackend
    if world > 1:
        torch.set_num_threads(1)
    ops.set_backend(EmuBackend())
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    T, nv = 6, 33

    class _Enc(torch.nn.Module):                     # pinned "encoder": a learnable map of the audio so gradients reach it
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(8, 768)
            # `layers` residual blocks under the names of the real encoder's Transformer layers: one gradient bucket each
            self.encoder = torch.nn.Module()
            self.encoder.layers = torch.nn.ModuleList(torch.nn.Linear(768, 768) for _ in range(layers))
            for lin in self.encoder.layers:
                torch.nn.init.normal_(lin.weight, 0, 0.01)

        def forward(self, wave, dataset, frame_num=None):
            f = wave[:, : frame_num * 8].reshape(wave.shape[0], frame_num, 8)
            h = self.proj(f)
            for l, lin in enumerate(self.encoder.layers):
                if skip_by_rank and (l == rank or (skip_by_rank == "rank0_all" and rank == 0)):
                    continue                         # LayerDrop: every rank skips a different layer; "rank0_all": rank 0 drops EVERY layer (none of its layer buckets' hooks fire)
                h = h + torch.tanh(lin(h))
            return types.SimpleNamespace(last_hidden_state=h)

    torch.manual_seed(21)
    model = a2f_amd.Faceformer(n_verts=nv, n_onehot=12, audio_encoder=_Enc())
    with torch.no_grad():
        model.vertice_map_r.weight.normal_(0, 0.1); model.vertice_map.weight.normal_(0, 0.01)
    model.train()
    model.decoder_regularisers = False               # the two runs must compute the same function
    step = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-3)
    g = torch.Generator().manual_seed(9)
    audio = torch.randn(S, T * 16000 // 60 + 8, generator=g)
    oh = torch.zeros(S, 12); oh[torch.arange(S), torch.arange(S) % 8] = 1
    templ = torch.randn(S, nv // 3, 3, generator=g)
    gt = templ.view(S, 1, nv // 3, 3) * 0.01 + 0.003 * torch.randn(S, T, nv // 3, 3, generator=g)
    cuts = [(S * r + world - 1) // world for r in range(world + 1)]      # as even as it gets: the last shards are one sequence shorter when world does not divide S
    sl = slice(cuts[rank], cuts[rank + 1])
    for _ in range(2):
        out = step(*TrainStep.unpack_batch({"audio": audio[sl], "one_hot": oh[sl], "verts": gt[sl], "template_vert": templ[sl]}))
    if world > 1 and layers:
        assert step.buckets is not None and len(step.buckets.bounds) == layers + 2                       # front | layers | decoder side
        assert step.buckets.launch_log == list(range(layers + 1, -1, -1)), step.buckets.launch_log       # the same descending order on every rank
    if rank == 0 or skip_by_rank:
        torch.save({"flat": step.opt.flat_p.clone(), "loss": float(out["loss"].detach())}, out_path if rank == 0 else out_path + f".rank{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_faceformer_dp2_bucketed_allreduce_matches_single_process():
    """The FaceFormer gradient exchange in buckets under the backward (trainer.BucketedAllReduce: front end | one bucket per encoder layer
    | decoder side, launched in descending order from autograd's post-accumulate hooks): two ranks against one process, and - with every
    rank skipping a DIFFERENT layer, as per-rank LayerDrop masks do - the same launch order on both ranks and identical parameters."""
    with tempfile.TemporaryDirectory() as d:
        p1, p2, p3 = os.path.join(d, "f1.pt"), os.path.join(d, "f2.pt"), os.path.join(d, "f3.pt")
        _run_ff(0, 1, 0, p1, 3)
        from a2f_amd import ops
        ops.set_backend(None)
        mp.spawn(_run_ff, args=(2, _free_port(), p2, 3), nprocs=2, join=True)
        a, b = torch.load(p1), torch.load(p2)
        assert float((a["flat"] - b["flat"]).norm() / a["flat"].norm()) < 1e-5
        mp.spawn(_run_ff, args=(2, _free_port(), p3, 3, True), nprocs=2, join=True)
        r0, r1 = torch.load(p3), torch.load(p3 + ".rank1")
        assert torch.equal(r0["flat"], r1["flat"])


@pytest.mark.parametrize("world,S", [(4, 8), (8, 8), (4, 7)])
def test_faceformer_bucketed_allreduce_world_4_and_8(world, S):
    """The bucketed exchange beyond two ranks (no multi-GPU hardware was available to any round: this is the N > 2 rehearsal): world 4 and 8 over gloo,
    every rank skipping a DIFFERENT encoder layer (per-rank LayerDrop patterns: its bucket's hooks never fire on that rank), even shards against one
    process with all sequences; uneven shards (7 sequences over 4 ranks: 2, 2, 2, 1): every rank launches the buckets in the same descending order
    (asserted inside the ranks) and ends on bit-identical parameters."""
    layers = world
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "f1.pt"), os.path.join(d, "f2.pt")
        from a2f_amd import ops
        if S % world == 0:
            _run_ff(0, 1, 0, p1, layers, False, S)
            ops.set_backend(None)
            mp.spawn(_run_ff, args=(world, _free_port(), p2, layers, False, S), nprocs=world, join=True)
            a, b = torch.load(p1), torch.load(p2)
            # Adam's first updates are lr * sign-like: an entry whose gradient is rounding noise takes either sign, and the summation order over 4 / 8
            # ranks differs from one process's - a handful of such entries move by 2 lr; everything else agrees to rounding
            diff = (a["flat"] - b["flat"]).abs()
            assert float(diff.norm() / a["flat"].norm()) < 1e-4
            assert float((diff > 1e-5).float().mean()) < 1e-3
        mp.spawn(_run_ff, args=(world, _free_port(), p2, layers, True, S), nprocs=world, join=True)
        r0 = torch.load(p2)
        for r in range(1, world):
            assert torch.equal(r0["flat"], torch.load(p2 + f".rank{r}")["flat"]), r
        if (world, S) == (8, 8):
            # the extreme LayerDrop pattern: rank 0 drops EVERY Transformer layer (on that rank only the front-end and the decoder-side bucket ever see a gradient), the
            # other ranks one each: rank 0 must still issue all ten collectives in the same descending order (asserted inside the ranks) or the job would hang / mis-sum
            mp.spawn(_run_ff, args=(world, _free_port(), p2, layers, "rank0_all", S), nprocs=world, join=True)
            r0 = torch.load(p2)
            for r in range(1, world):
                assert torch.equal(r0["flat"], torch.load(p2 + f".rank{r}")["flat"]), r


def test_trainstep_with_use_graph_drops_the_bucket_hooks():
    """a captured step cannot run autograd hooks that launch collectives - TrainStep(use_graph=True) must fall back to the one blocking
    all-reduce inside FlatAdam.step() and leave no hook behind; close() does the same for an eager step."""
    import a2f_amd
    from a2f_amd.trainer import TrainStep
    os.environ["A2F_DP_FORCE_COLLECTIVES"] = "1"
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
        model = a2f_amd.Faceformer(n_verts=33, n_onehot=12, audio_encoder=torch.nn.Linear(8, 768))
        n_hooks = lambda: sum(len(getattr(p, "_post_accumulate_grad_hooks", None) or {}) for p in model.parameters())
        eager = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-3)
        assert eager.buckets is not None and n_hooks() > 0
        eager.close()
        assert eager.buckets is None and n_hooks() == 0
        graph = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-3, use_graph=True)
        assert graph.buckets is None and n_hooks() == 0
    finally:
        del os.environ["A2F_DP_FORCE_COLLECTIVES"]
        if dist.is_initialized():
            dist.destroy_process_group()


def test_faceformer_dp2_gloo_matches_single_process():
    """Two ranks, two sequences each, against one process with all four: the mean over equal shards of per-sequence means is the
    mean over all sequences, so after two steps (all-reduce of the flat gradient buffer incl. the encoder stand-in's parameters,
    fused Adam with the 1/world factor) both land on the same parameters."""
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "f1.pt"), os.path.join(d, "f2.pt")
        _run_ff(0, 1, 0, p1)
        from a2f_amd import ops
        ops.set_backend(None)
        mp.spawn(_run_ff, args=(2, _free_port(), p2), nprocs=2, join=True)
        a, b = torch.load(p1), torch.load(p2)
        rel = float((a["flat"] - b["flat"]).norm() / a["flat"].norm())
        assert rel < 1e-5, rel
