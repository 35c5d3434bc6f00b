#!/usr/bin/env python3
"""bench.py - mesh-frames/s of full training steps (fwd + loss + bwd + [all-reduce] + Adam) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  With N > 1 and no torchrun environment the script
starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a CHILD process before anything touches the
GPU, relays the child's JSON line and exits with its return code; under torchrun it is one rank per GPU (RCCL).
Rank 0 prints ONE JSON line.

Headline (``metric``/``value``) = BASELINE.json configs[1]: VOCA (DeepSpeech-29 features, 8-subject one-hot), synthetic
VOCASET batch 32 sequences x 600 frames = 19 200 frame windows (29,16) per GPU per step, 5023-vertex meshes, weak scaling
(every rank runs the same per-GPU batch on its own shard, seed 1234+rank; the only data-path exchange is the gradient
all-reduce).  ``secondary`` carries the other two model families on their BASELINE configurations, same step definition:
  * FaceFormer (configs[2]): wav2vec2-base encoder + autoregressive decoder, 64 sequences x 600 frames per GPU per step, fp32 storage
  * Audio2Mesh: B = 1024 windows (52,32) per GPU per step
each with its own ``roofline`` and ``cpu_baseline``.  ``--workload faceformer|audio2mesh`` makes one of them the headline.

Extra objects:
  roofline     - the dominant HIP kernel's ALGORITHMIC bytes (or flops) per launch / its average launch duration, timed
                 live with HIP events recorded on the launch stream over a separate, identically shaped pass (so event
                 overhead never sits inside the headline timing); ``measured_stream_GBs`` = a float4 copy / triad kernel of
                 this library timed in the same job (what the box gives plain streaming kernels).
  cpu_baseline - the CPU oracle (oracle/, "port") timed on this node's host cores on a bounded sample of the same
                 workload (rank 0, N=1 only); ``host_cores`` = os.cpu_count(), ``cores`` = threads actually used.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NV3 = 15069
FRAME_BYTES = NV3 * 4             # one fp32 mesh = 60 276 B

# algorithmic HBM bytes per frame of each tagged kernel (DESIGN.md "Kernels and rooflines")
ALGO_BYTES_PER_FRAME = {
    "head_fused": 3 * FRAME_BYTES + 50 * 4 * 2,      # template + gt read, pred written, H read, dH written
    "head_fused_ff": 2 * FRAME_BYTES + 64 * 4 * 2,   # FaceFormer: gt read, pred written (template amortised over the sequence)
}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3          # exact-fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0        # dense bf16 matrix peak
FF_MFLOP_PER_FRAME = 857.0        # 3 x 2 x (141.9 M encoder + 1.0 M decoder/head) MAC at T = 600
A2M_MFLOP_PER_FRAME = 393.0       # 3 x 2 x 65 509 514 MAC
FF_LAYER_MFLOP_PER_FRAME = 6.0 * (7.078 + 0.92) # one of the 12 Transformer layers: 3 x 2 x (7.08 M MAC of projections / FFN + 0.92 M MAC of attention at T = 600)
PMC_TRAFFIC_FILE = "r05_pmc_traffic.json"   # this round's PMC measurement of the dominant kernels (tools/pmc_traffic.sh)


# ------------------------------------------------------------------------------------------------ self launch (N > 1)
def self_launch(argv, n: int) -> int:
    """Start the N-rank job as a child torchrun and relay its output.  Runs BEFORE any GPU call of this process (a process
    that has initialised the GPU must not exec/replace itself; a child process is fine)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return r.returncode if lines or r.returncode else 1


# ------------------------------------------------------------------------------------------------ synthetic batches
def make_voca_batch(n_seq: int, n_frames: int, rank: int, device, features: bool = False):
    """C2 of SURVEY §8(d): features (n_seq, n_frames, 29) ~ N(0,1) -> 16-frame windows (zero-padded ends), one subject
    per sequence, per-subject template ~ N(0,5), gt = template + N(0,0.5) (already x100-scaled units)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    feats = torch.randn(n_seq, n_frames, 29, generator=g)
    from a2f_amd.data import feature_windows
    x = feature_windows(feats)                                              # (S*T, 29, 16): window t = frames t-8..t+7, zero-padded ends
    subj = torch.randint(0, 8, (n_seq,), generator=g)
    one_hot = torch.zeros(n_seq, 12)
    one_hot[torch.arange(n_seq), subj] = 1.0
    one_hot = one_hot.repeat_interleave(n_frames, 0)
    templates = torch.randn(8, 5023, 3, generator=g) * 5.0
    B = n_seq * n_frames
    x, one_hot = x.to(device), one_hot.to(device)
    templ = templates.to(device)[subj.to(device)].repeat_interleave(n_frames, 0).contiguous()     # (B,5023,3)
    gg = torch.Generator(device=device).manual_seed(4321 + rank)
    gt = templ + torch.randn(B, 5023, 3, generator=gg, device=device) * 0.5
    return (feats.to(device) if features else x), one_hot, gt, templ


def make_ff_batch(n_seq: int, T: int, rank: int, device):
    """C3 of SURVEY §8(d): audio (S, N) ~ N(0,0.1) at 16 kHz with N*60//16000 == T, one subject per sequence, gt (S,T,5023,3)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    n = T * 16000 // 60 + 8
    assert n * 60 // 16000 == T
    audio = (torch.randn(n_seq, n, generator=g) * 0.1).to(device)
    subj = torch.randint(0, 8, (n_seq,), generator=g)
    one_hot = torch.zeros(n_seq, 12)
    one_hot[torch.arange(n_seq), subj] = 1.0
    templ = (torch.randn(n_seq, 5023, 3, generator=g) * 5.0).to(device)
    gg = torch.Generator(device=device).manual_seed(4321 + rank)
    gt = templ.view(n_seq, 1, 5023, 3) + torch.randn(n_seq, T, 5023, 3, generator=gg, device=device) * 0.5
    return audio, one_hot.to(device), gt, templ


def make_a2m_batch(B: int, rank: int, device):
    """Audio2Mesh windows (B,52,32) ~ N(0,1) (the MFCC front end's output shape, config.yaml), 12-way one-hot."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.randn(B, 52, 32, generator=g).to(device)
    one_hot = torch.zeros(B, 12)
    one_hot[torch.arange(B), torch.randint(0, 12, (B,), generator=g)] = 1.0
    templ = (torch.randn(B, 5023, 3, generator=g) * 5.0).to(device)
    gg = torch.Generator(device=device).manual_seed(4321 + rank)
    gt = templ + torch.randn(B, 5023, 3, generator=gg, device=device) * 0.5
    return x, one_hot.to(device), gt, templ


# ------------------------------------------------------------------------------------------------ CPU baselines (oracle)
CPU_BASELINE_THREADS = 16      # ONE policy for every CPU leg: the host-core share a one-GPU box of this pool grants its job (`cores` in each cpu_baseline)


def _pick_threads(step):
    """The CPU legs all run on the same thread count: the 16-core share of a one-GPU box (a GPU box exposes every host thread - 256 - but
    grants 16; torch's default would oversubscribe it).  A2F_CPU_BASELINE_THREADS overrides."""
    import torch
    nt = max(1, min(int(os.environ.get("A2F_CPU_BASELINE_THREADS", CPU_BASELINE_THREADS)), os.cpu_count() or 1))
    torch.set_num_threads(nt)
    step()                                  # warm-up (allocator, thread pool)
    return nt


def _time_cpu(step, seconds, min_steps=2):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or n < min_steps:
        step(); n += 1
    return n, time.perf_counter() - t0


def cpu_baseline_voca(seconds: float = 10.0, B: int = 2048):
    """Oracle VOCA train step (fwd + VocaLoss + bwd + torch Adam) on the host cores, bounded sample."""
    import torch
    from oracle import losses as o_loss, voca as o_voca, weights as W
    g = torch.Generator().manual_seed(1234)
    sd = W.formula_state_dict(W.voca_shapes(NV3), gains={"decoder.4": 4.0})
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-4, weight_decay=1e-5)
    x = torch.randn(B, 29, 16, generator=g)
    oh = torch.zeros(B, 12); oh[:, 3] = 1
    t = torch.randn(B, 5023, 3, generator=g) * 5
    gt = t + torch.randn(B, 5023, 3, generator=g) * 0.5

    def step():
        opt.zero_grad(set_to_none=True)
        l = o_loss.voca_loss(o_voca.voca_forward(P, x, oh, t), gt)
        l["loss"].backward()
        opt.step()
    nt = _pick_threads(step)
    n, dt = _time_cpu(step, seconds, 3)
    return {"value": round(B * n / dt, 1), "unit": "mesh-frames/s", "cores": nt, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle VOCA train step (fwd+VocaLoss+bwd+Adam), B={B} of the step's 19200 windows x {n} steps in {dt:.1f}s, fp32 torch CPU"}


def cpu_baseline_a2m(seconds: float = 8.0, B: int = 128):
    import torch
    from oracle import audio2mesh as o_a2m, losses as o_loss, weights as W
    g = torch.Generator().manual_seed(1234)
    sd = W.formula_state_dict(W.audio2mesh_shapes(NV3, 12), gains={"output_net.4": 4.0})
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    opt = torch.optim.Adam([v for v in P.values() if v.requires_grad], lr=1e-4, weight_decay=1e-5)
    x = torch.randn(B, 52, 32, generator=g)
    oh = torch.zeros(B, 12); oh[:, 3] = 1
    t = torch.randn(B, 5023, 3, generator=g) * 5
    gt = t + torch.randn(B, 5023, 3, generator=g) * 0.5

    def step():
        opt.zero_grad(set_to_none=True)
        l = o_loss.voca_loss(o_a2m.audio2mesh_forward(P, x, oh, t, training=True), gt)
        l["loss"].backward()
        opt.step()
    nt = _pick_threads(step)
    n, dt = _time_cpu(step, seconds, 2)
    return {"value": round(B * n / dt, 1), "unit": "mesh-frames/s", "cores": nt, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle Audio2Mesh train step (train-mode BN, fwd+VocaLoss+bwd+Adam), B={B} windows x {n} steps in {dt:.1f}s, fp32 torch CPU"}


def cpu_baseline_faceformer(seconds: float = 10.0, T_inc: int = 600, T_ref: int = 200):
    """Oracle FaceFormer train step in the two forms BASELINE.md §4 names: the reference-shaped O(T^3) loop
    (faceformer.py:154-185 as written; bounded to T_ref frames - its autograd graph holds T(T+1)/2 x 60 KB) and the
    incremental K/V-cache form the HIP path evaluates, both behind the wav2vec2-base encoder, + FaceFormerLoss + Adam."""
    import torch
    from oracle import faceformer as o_ff, losses as o_loss, wav2vec2 as o_w2v, weights as W
    sd = W.formula_state_dict(W.faceformer_decoder_shapes(NV3, 12, 64),
                              gains={"vertice_map_r.weight": 4.0, "vertice_map.weight": 0.05, "obj_vector": 2.0, "in_proj_weight": 1.5})
    esd = W.formula_state_dict(W.wav2vec2_shapes(), gains={"original0": 40.0})
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.startswith("PPE")}
    E = {k: v.clone().requires_grad_(True) for k, v in esd.items()}
    opt = torch.optim.Adam(list(P.values()) + list(E.values()), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(1234)
    oh = torch.zeros(1, 12); oh[0, 3] = 1
    templ = torch.randn(1, 5023, 3, generator=g) * 5
    out = {}
    for form, T in (("incremental", T_inc), ("reference_loop", T_ref)):
        audio = torch.randn(1, T * 16000 // 60 + 8, generator=g) * 0.1
        gt = templ.view(1, 1, 5023, 3) + torch.randn(1, T, 5023, 3, generator=g) * 0.5

        def step():
            opt.zero_grad(set_to_none=True)
            enc = o_w2v.wav2vec2_forward(E, o_w2v.normalize_waveform(audio), T)
            dec = o_ff.decode_incremental if form == "incremental" else o_ff.decode_reference_loop
            pred = dec(P, enc, oh, templ)
            l = o_loss.faceformer_loss(pred, gt)
            l["loss"].backward()
            opt.step()
        if form == "incremental":
            nt = _pick_threads(step)
        n, dt = _time_cpu(step, seconds if form == "incremental" else seconds / 2, 1)
        out[form] = {"value": round(T * n / dt, 1), "T": T, "steps": n, "seconds": round(dt, 1)}
    return {"value": out["incremental"]["value"], "unit": "mesh-frames/s", "cores": nt, "host_cores": os.cpu_count(), "kind": "port",
            "reference_shaped_loop": out["reference_loop"],
            "sample": f"oracle FaceFormer train step (wav2vec2-base fwd+bwd, decoder, FaceFormerLoss, Adam over 96.4 M parameters), one sequence: "
                      f"incremental K/V-cache form T={T_inc} x {out['incremental']['steps']} steps in {out['incremental']['seconds']}s (= value); "
                      f"reference-shaped O(T^3) loop bounded to T={T_ref} (T=600 would retain 10.9 GB of activations): "
                      f"{out['reference_loop']['value']} frames/s; fp32 torch CPU"}


# ------------------------------------------------------------------------------------------------ measured stream rate
def measured_stream_rates(device):
    """Copy (1R+1W) and triad (2R+1W) GB/s of this library's float4 streaming kernels on 1 GiB buffers, HIP events."""
    import torch
    from a2f_amd import ops
    be = ops.backend()
    n = 1 << 28                                              # 2^28 floats = 1 GiB per buffer (beyond the 256 MiB Infinity Cache)
    a = torch.ones(n, device=device); b = torch.ones(n, device=device); c = torch.empty(n, device=device)
    res = {}
    for name, fn, nbuf in (("copy", lambda: be.stream_copy(c, a), 2), ("triad", lambda: be.stream_triad(c, a, b), 3)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = round(nbuf * n * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b, c
    return res


# ------------------------------------------------------------------------------------------------ timed regions
def timed(step_fn, steps, warmup, barrier, world, device, budget_s=8.0, max_blocks=9):
    """W untimed warm-up steps, then BLOCKS of exactly ``steps`` steps, each bracketed by barrier + synchronize on both sides and reduced
    with MAX over the ranks.  The reported time is the MEDIAN block (>= 50 timed steps, median): the number of blocks is the
    smallest that reaches 50 timed steps (at least 3, at most ``max_blocks``), cut so that the timed region stays within ``budget_s``
    seconds (every rank takes the decision from the same max-reduced first block).  Returns (median block seconds, last loss, info)."""
    import torch
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        out = step_fn()

    def one_block():
        nonlocal out
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step_fn()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return el

    blocks = [one_block()]
    want = max(3, min(max_blocks, -(-50 // steps)))
    want = max(1, min(want, int(budget_s / max(blocks[0], 1e-9))))
    while len(blocks) < want:
        blocks.append(one_block())
    loss_val = float(out["loss"].detach())
    assert loss_val == loss_val, "loss is NaN"
    srt = sorted(blocks)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    info = {"blocks": len(blocks), "steps_per_block": steps, "statistic": "median block, max over ranks per block",
            "ms_per_step_per_block": [round(1e3 * b / steps, 4) for b in blocks]}
    return med, loss_val, info


def gemm_prof_summary(be):
    """(TFLOP/s, launches, total ms) of the tagged MFMA GEMM launches recorded in be.prof (HIP events on the launch stream)."""
    evs = be.prof.get("gemm_mfma", []) if be.prof else []
    fl = getattr(be, "prof_flops", {}).get("gemm_mfma", 0.0)
    ms = sum(a.elapsed_time(b) for a, b in evs)
    return (fl / (ms * 1e-3) / 1e12 if ms > 0 else None), len(evs), ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="voca", choices=["voca", "faceformer", "audio2mesh", "faceformer_long"])
    ap.add_argument("--ff-long-seqs", type=int, default=12)
    ap.add_argument("--ff-long-frames", type=int, default=3000)
    ap.add_argument("--seqs", type=int, default=32)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--ff-seqs", type=int, default=64)
    ap.add_argument("--ff-frames", type=int, default=600)
    ap.add_argument("--a2m-batch", type=int, default=1024)
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--encoder-precision", default="split", choices=["split", "bf16"],
                    help="FaceFormer encoder products: split = bf16x3 (fp32-class, the parity setting); bf16 = plain bf16 operands "
                         "(the reference's `percision: 16-mixed`)")
    ap.add_argument("--input", default="windows", choices=["windows", "features"],
                    help="windows: x = (B,29,16) as the reference's Voca.forward takes it (default, the headline); features: x = (S,T,29), "
                         "the 16-frame windows are built inside the trunk kernel (SURVEY §8(f) N4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--ff-deterministic", action="store_true", help="FaceFormer: switch the train-mode regularisers off (A/B timing)")
    ap.add_argument("--no-batch-points", action="store_true", help="FaceFormer: skip the 1- and 16-sequence points (clean kernel profiles of the main step)")
    ap.add_argument("--no-graph", action="store_true", help="launch the ~90 kernels of a step eagerly instead of replaying a hipGraph")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))      # child torchrun; nothing in THIS process has touched the GPU

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # A2F_BENCH_BACKEND=gloo A2F_BENCH_SHARE_GPU=1: rehearsal of the multi-rank flow on a one-GPU box (every rank on cuda:0,
    # collectives through gloo); the real runs use one GPU per rank and RCCL
    backend = os.environ.get("A2F_BENCH_BACKEND", "nccl")
    if os.environ.get("A2F_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("A2F_BENCH_FORCE_DIST") == "1"      # rehearsal: a one-rank RCCL group (tests/test_gpu_models.py)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import a2f_amd
    from a2f_amd import ops
    from a2f_amd.trainer import TrainStep

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dist_info = None
    if world > 1 or force_dist:
        # evidence for the scaling run that the collective library really saw N ranks on N devices: an all-reduce whose result every
        # rank checks (sum of rank + 1 = N (N + 1) / 2) and the device each rank is bound to
        chk = torch.tensor([float(rank + 1)], device=device)
        dist.all_reduce(chk)
        want_sum = world * (world + 1) / 2
        assert float(chk.item()) == want_sum, f"all-reduce self-check failed on rank {rank}: {float(chk.item())} != {want_sum}"
        prop = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}", "name": prop.name,
                "uuid": str(getattr(prop, "uuid", "")), "pci_bus_id": getattr(prop, "pci_bus_id", None)}
        devs = [None] * world
        dist.all_gather_object(devs, mine)
        dist_info = {"backend": dist.get_backend(), "collective_library": "RCCL" if backend == "nccl" else backend, "rccl_world": dist.get_world_size(),
                     "allreduce_selfcheck": f"sum(rank + 1) = {float(chk.item()):.0f} on every rank (expected {want_sum:.0f})",
                     "distinct_devices": len({(d["uuid"], d["pci_bus_id"], d["device"]) for d in devs}), "ranks": devs}

    be = ops.backend()
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    want_roof = rank == 0 and not args.no_roofline
    stream_rates = measured_stream_rates(device) if want_roof else None

    # ------------------------------------------------------------------ VOCA (BASELINE configs[1])
    def run_voca(steps, warmup):
        torch.manual_seed(1234)                                  # identical initial weights on every rank
        model = a2f_amd.Voca(n_verts=NV3, n_onehot=12).to(device)
        step = TrainStep(model, a2f_amd.VocaLoss(), lr=1e-4, use_graph=not args.no_graph)
        x, one_hot, gt, templ = make_voca_batch(args.seqs, args.frames, rank, device, features=args.input == "features")
        B = one_hot.shape[0]
        elapsed, loss_val, tinfo = timed(lambda: step(x, one_hot, gt, templ), steps, max(warmup, 3), barrier, world, device)   # >= 3: two eager warm-up steps + the capture step
        res = {
            "value": round(world * B * steps / elapsed, 1), "unit": "mesh-frames/s", "steps": steps,
            "ms_per_step": round(1e3 * elapsed / steps, 4), "timing": tinfo,
            "dtype": "f32 (head matrix products: bf16x3 split operands, f32 accumulate)",
            "config": {"workload": f"VOCA train step (fwd+VocaLoss+bwd+Adam), {args.seqs} seq x {args.frames} frames = {B} windows "
                                   f"(29,16) per GPU per step{' built on chip from (S,T,29) features' if args.input == 'features' else ''}, 5023-vertex meshes, BASELINE configs[1]",
                       "frames_per_gpu_step": B, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6),
                       "launch": ("eager" if not step.use_graph else "hipGraph replay (fwd+loss+bwd) + eager all-reduce/Adam")
                                 + ("; head-gradient all-reduce overlapped with the trunk backward" if step.overlap else "")},
        }
        if want_roof:
            be.profile_head(True)                                  # HIP events on the launch stream around head_fused_kernel alone
            nprof = max(3, min(steps, 60))                         # the in-library ring holds 64 event pairs
            eager = TrainStep.__new__(TrainStep)                   # same model/optimizer, eager launches (events are not graph nodes)
            eager.__dict__.update(step.__dict__)
            eager.use_graph = False
            for _ in range(nprof):
                eager._fwd_bwd(x, one_hot, gt, templ)              # forward + loss + backward only: rank 0 runs this leg alone, so no collective
            torch.cuda.synchronize()
            head_ms, nl = be.profile_head_ms()
            be.profile_head(False)
            dom = "head_x3_kernel"
            bytes_per_launch = ALGO_BYTES_PER_FRAME["head_fused"] * B
            achieved = bytes_per_launch / (head_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes: tools/pmc_head.sh writes the
            # file; a PMC pass cannot run inside this process).  Only a measurement of THIS round's kernel at THIS shape is reported.
            traffic, traffic_src = None, None
            tj = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
            if os.path.exists(tj):
                ent = json.load(open(tj)).get(dom)
                if isinstance(ent, dict) and ent.get("frames") == B:
                    traffic, traffic_src = ent.get("hbm_bytes_per_launch"), f"profiles/{PMC_TRAFFIC_FILE} ({ent.get('how', 'rocprofv3 --pmc')})"
            res["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(head_ms, 4), "launches_timed": nl,
                               "measured_stream_GBs": stream_rates,
                               "whole_step_frac": round(B * steps / elapsed * 182732 / 1e9 / HBM_PEAK_GBS, 4)}
        stage("voca: timed blocks + head roofline done")
        if rank == 0 and world == 1 and not args.no_secondary:
            # LABELLED VARIANT, not the headline: the same step with no prediction written (Voca.store_pred = False).  The reference's training_step never
            # reads pred (mse_error = rec_loss / 3 comes from the loss sums), so a trainer may drop the store; the Module API (forward returns pred) is what
            # the headline keeps.  Own algorithmic bytes: 182 732 - 60 276 = 122 456 per frame.
            try:
                model.store_pred = False
                vstep = TrainStep(model, a2f_amd.VocaLoss(), lr=1e-4, use_graph=not args.no_graph)
                el, _lv, _ti = timed(lambda: vstep(x, one_hot, gt, templ), steps, 3, barrier, world, device, budget_s=3.0)
                res["variants"] = [{"variant": "store_pred=False (training step that reads the loss only; pred is not materialised)", "ms_per_step": round(1e3 * el / steps, 4),
                                    "value": round(B * steps / el, 1), "algorithmic_bytes_per_frame": 122456,
                                    "whole_step_frac": round(B * steps / el * 122456 / 1e9 / HBM_PEAK_GBS, 4)}]
                del vstep
            except Exception as e:
                res["variants"] = [{"variant": "store_pred=False", "error": repr(e)[:200]}]
            finally:
                model.store_pred = True
        if want_cpu:
            res["cpu_baseline"] = cpu_baseline_voca()
        del step, model
        return res

    # ------------------------------------------------------------------ FaceFormer (BASELINE configs[2])
    def ff_mflop_per_frame(T, layers_run=12.0):
        """Algorithmic fp32 MFLOP per frame of a FaceFormer train step (3 x 2 x MAC; SURVEY 8(d): 857 at T = 600 with all 12 layers): conv
        feature encoder 41.0 M MAC, positional conv 4.7, decoder + head 1.0, per Transformer layer 7.08 (projections + FFN) + 0.92 T / 600
        (attention is quadratic in T); ``layers_run`` = the mean number of layers LayerDrop left in the timed steps."""
        return 6.0 * (41.0 + 4.7 + 1.0 + 0.22 + layers_run * (7.078 + 0.92 * T / 600.0))

    def ff_model():
        model = a2f_amd.Faceformer(n_verts=NV3, n_onehot=12).to(device)
        with torch.no_grad():                                    # the reference zero-initialises both vertice maps (faceformer.py:132-135):
            model.vertice_map_r.weight.normal_(0, 0.1)           # non-zero values so the recurrence and its backward do real work
            model.vertice_map.weight.normal_(0, 0.001)
        if args.encoder_precision == "bf16":
            model.encoder_autocast = torch.bfloat16
        model.train()                                            # Lightning's fit puts the reference's module in train mode
        if args.ff_deterministic:                                # A/B: the deterministic (eval-mode) function under the same step
            model.audio_encoder.train_regularisers = model.decoder_regularisers = False
        return model

    def run_faceformer(steps, warmup, S=None, T=None, config_name="BASELINE configs[2]", cpu=True, batch_points=(1, 16), long=False):
        torch.manual_seed(1234)
        S, T = S or args.ff_seqs, T or args.ff_frames
        model = ff_model()
        torch.manual_seed(4321 + 7919 * rank)                    # same initial weights on every rank (seed above), different dropout masks per rank
        step = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-4)
        audio, one_hot, gt, templ = make_ff_batch(S, T, rank, device)
        enc = model.audio_encoder
        enc.layers_run, enc.forwards_run = 0, 0
        elapsed, loss_val, tinfo = timed(lambda: step(audio, one_hot, gt, templ), steps, max(warmup, 2), barrier, world, device,
                                         budget_s=12.0 if not long else 4.0, max_blocks=5 if not long else 3)
        stage(f"faceformer {S} x {T}: timed blocks done")
        layers_run = enc.layers_run / max(enc.forwards_run, 1) if getattr(enc, "forwards_run", 0) else 12.0
        fps = world * S * T * steps / elapsed
        enc_path = getattr(model.audio_encoder, "compute_path", "torch ops (rocBLAS / MIOpen / SDPA)")
        split = args.encoder_precision != "bf16"
        res = {
            "value": round(fps, 1), "unit": "mesh-frames/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3), "timing": tinfo,
            "dtype": "f32 storage; " + ("encoder products (GEMMs and attention, forward and backward) as plain bf16 MFMA (16-mixed), fp32 accumulate / softmax / LayerNorm"
                                        if not split else "encoder products (GEMMs and attention, forward and backward) as bf16x3 split MFMA (fp32-class: the parity "
                                        "setting, <= 1e-4 vs the oracle), fp32 accumulate / softmax / LayerNorm"),
            "config": {"workload": f"FaceFormer train step (wav2vec2-base encoder + incremental decoder + fused head/FaceFormerLoss, bwd, Adam over 96.4 M "
                                   f"parameters), {S} independent batch-1 sequences x {T} frames per GPU per step, {config_name}",
                       "frames_per_gpu_step": S * T, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6), "encoder": enc_path,
                       "train_mode": ("deterministic function (regularisers off, --ff-deterministic)" if args.ff_deterministic else
                                      "reference train mode: wav2vec2 dropout x6 kinds of sites / LayerDrop 0.1 / SpecAugment (exactly the reference's); decoder + PPE "
                                      "dropout 0.1 with ONE mask per frame row - a deliberate deviation from the reference's per-loop-step redraw "
                                      "(faceformer.py:154-185; DESIGN 3.6, fixture G14); masks drawn on the device every step"),
                       "mean_transformer_layers_run": round(layers_run, 3),
                       "gradient_exchange": ("none (one rank)" if world == 1 and step.buckets is None else
                                             f"{len(step.buckets.bounds)} buckets (encoder front end | Transformer layers | decoder side), async all-reduce "
                                             "launched from autograd hooks as the backward passes each bucket" if step.buckets is not None
                                             else "one blocking all-reduce of the flat gradient buffer after the backward")},
        }
        if want_roof:
            # whole step and the dominant GEMM variant against the pipe the kernels run on: the DENSE bf16 matrix peak.  A bf16x3 kernel
            # executes three bf16 MFMA flops per algorithmic (fp32) flop: `achieved` = executed, `algorithmic` = 2 M N K / t
            mult = 3.0 if split else 1.0
            mf = ff_mflop_per_frame(T, layers_run)
            alg = fps / world * mf * 1e6 / 1e12
            roof = {"bound": "mfma", "achieved": round(mult * alg, 2), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(mult * alg / MFMA_BF16_PEAK_TF, 4),
                    "traffic": None, "algorithmic": round(alg, 2), "algorithmic_frac": round(alg / MFMA_BF16_PEAK_TF, 4), "executed_per_algorithmic": mult,
                    "algorithmic_mflop_per_frame": round(mf, 1),
                    "note": "whole step: algorithmic fp32 flops (3 x 2 x MAC per frame, LayerDrop-skipped layers subtracted) / step time; achieved = executed "
                            "bf16 MFMA flops (x3 for split products) against the dense bf16 peak (2.5 PFLOP/s)"}
            # per-launch HIP events are only meaningful when nothing else shares the GPU: this pass runs the same kernels with the
            # two-stream half-batch pipeline and the weight-gradient lane switched off (one stream, launches back to back)
            from a2f_amd import encoder_ops
            old_lane, old_pipe, old_buckets = encoder_ops.WGRAD_STREAM, step.pipeline, step.buckets
            encoder_ops.WGRAD_STREAM, step.pipeline, step.buckets = False, False, None       # buckets off: rank 0 runs this leg alone, no collective may start
            be.prof, be.prof_flops, be.prof_shapes = {}, {}, {}
            for _ in range(2):
                step._fwd_bwd(audio, one_hot, gt, templ)         # forward + loss + backward only (no gradient exchange, no Adam)
            torch.cuda.synchronize()
            encoder_ops.WGRAD_STREAM, step.pipeline, step.buckets = old_lane, old_pipe, old_buckets
            tf, nl, ms = gemm_prof_summary(be)
            if tf is not None:
                roof["gemm_family"] = {"name": "gemm_mfma_kernel (all encoder / decoder-side GEMM launches of 2 steps, single stream)", "launches_timed": nl,
                                       "total_ms": round(ms, 3), "algorithmic_TFLOPs": round(tf, 2), "achieved_TFLOPs": round(mult * tf, 2),
                                       "frac": round(mult * tf / MFMA_BF16_PEAK_TF, 4)}
            top = None
            for key, ent in be.prof_shapes.items():
                tms = sum(a.elapsed_time(b) for a, b in ent["events"])
                if top is None or tms > top[1]:
                    top = (key, tms, ent)
            if top is not None:
                (lay, M_, N_, K_, bt_, win, prec, img), tms, ent = top
                n = len(ent["events"])
                ktf = ent["flops"] * n / (tms * 1e-3) / 1e12
                kmult = {0: 1.0, 1: 3.0, 2: 6.0}[prec]
                traffic = None
                tj = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
                if os.path.exists(tj):
                    e2 = json.load(open(tj)).get(f"gemm_mfma_kernel {lay} {M_}x{N_}x{K_}")
                    if isinstance(e2, dict):
                        traffic = e2.get("hbm_bytes_per_launch")
                roof["kernel"] = {"name": f"gemm_mfma_kernel {lay} M={M_} N={N_} K={K_}" + (f" batch={bt_}" if bt_ > 1 else "") + (" windowed (conv)" if win else "")
                                          + (" image operands" if img else "") + f" precision={prec}",
                                  "launches_timed": n, "avg_launch_ms": round(tms / n, 4), "total_ms": round(tms, 3),
                                  "algorithmic_TFLOPs": round(ktf, 2), "achieved": round(kmult * ktf, 2), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                  "frac": round(kmult * ktf / MFMA_BF16_PEAK_TF, 4), "traffic": traffic,
                                  "note": "the launch variant (layout, shape) with the largest total time in the single-stream pass"}
            hev = be.prof.get("head_fused", [])
            if hev:
                hms = sum(a.elapsed_time(b) for a, b in hev) / len(hev)
                hb = ALGO_BYTES_PER_FRAME["head_fused_ff"] * (2 * S * T / len(hev))          # frames per launch over the 2 profiled steps (single stream)
                roof["head_hbm"] = {"kernel": "head_x3_kernel<64> + fold", "achieved": round(hb / (hms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(hb / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(hms, 4), "launches_timed": len(hev)}
            be.prof = be.prof_shapes = None
            res["roofline"] = roof
        stage(f"faceformer {S} x {T}: roofline pass done")
        # the reference trains ONE sequence per optimizer step (train.py:33): the same step at S = 1 and S = 16 beside the S-sequence figure
        pts = []
        if args.no_batch_points:
            batch_points = ()
        for s_pt in batch_points:
            if s_pt >= S:
                continue
            a2, o2, g2, t2 = audio[:s_pt], one_hot[:s_pt], gt[:s_pt], templ[:s_pt]
            el, _lv, _ti = timed(lambda: step(a2, o2, g2, t2), 3, 2, barrier, world, device, budget_s=2.0, max_blocks=3)
            pts.append({"seqs_per_step": s_pt, "ms_per_step": round(1e3 * el / 3, 3), "value": round(world * s_pt * T * 3 / el, 1)})
        if 1 in batch_points and S > 1 and not args.no_graph:
            # the reference's own step (batch = 1) as ONE hipGraph replay: ~2 500 eager launches are host-bound at one sequence per step; the captured step
            # keeps its random state on the device (dropout.StepRandomness: mask seeds, LayerDrop decisions, SpecAugment rows refreshed before every replay)
            try:
                gstep = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-4, use_graph=True)
                a2, o2, g2, t2 = audio[:1].clone(), one_hot[:1].clone(), gt[:1].clone(), templ[:1].clone()
                for _ in range(3):
                    gstep(a2, o2, g2, t2)                                  # two eager warm-ups, then the capture
                el, _lv, _ti = timed(lambda: gstep(a2, o2, g2, t2), 10, 2, barrier, world, device, budget_s=2.0, max_blocks=3)
                for p_ in pts:
                    if p_["seqs_per_step"] == 1:
                        # measured: the replay is NOT faster than the eager step (22.0 vs 21.4 ms) - one sequence per step is bound by the GPU's own serial chain
                        # (595 kernels: the 600-step decoder recurrence 7.3 ms on one CU, 600-row GEMMs 6.5 ms), not by the host; the eager figure stays the point
                        p_["graph_replay_ms_per_step"] = round(1e3 * el / 10, 3) if gstep.use_graph else None
                del gstep
            except Exception as e:                                         # the eager figure stands
                for p_ in pts:
                    if p_["seqs_per_step"] == 1:
                        p_["graph_error"] = repr(e)[:200]
        stage(f"faceformer {S} x {T}: batch points done")
        if pts:
            res["sequences_per_step_points"] = pts
        if want_cpu and cpu:
            res["cpu_baseline"] = cpu_baseline_faceformer()
        del step
        return res, model, (audio, one_hot, templ)

    def run_faceformer_entry(steps, warmup):
        res, model, _ = run_faceformer(steps, warmup)
        del model
        return res

    # ------------------------------------------------------------------ Audio2Mesh
    def run_audio2mesh(steps, warmup):
        torch.manual_seed(1234)
        B = args.a2m_batch
        model = a2f_amd.Audio2Mesh(n_verts=NV3, n_onehot=12).to(device).train()
        step = TrainStep(model, a2f_amd.VocaLoss(), lr=1e-4, use_graph=not args.no_graph)
        x, one_hot, gt, templ = make_a2m_batch(B, rank, device)
        elapsed, loss_val, tinfo = timed(lambda: step(x, one_hot, gt, templ), steps, max(warmup, 3), barrier, world, device)
        fps = world * B * steps / elapsed
        res = {
            "value": round(fps, 1), "unit": "mesh-frames/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3), "timing": tinfo,
            "dtype": "f32 storage and accumulate; convolution products as split-bf16 MFMA: forward bf16x6 (fp32-class: its rounding feeds the BatchNorm "
                     "statistics), weight / data gradients bf16x3 (<= 1e-4 on every gradient of the reference fixture G2)",
            "config": {"workload": f"Audio2Mesh train step (train-mode BatchNorm, fwd+VocaLoss+bwd+Adam), B = {B} windows (52,32) per GPU per step",
                       "frames_per_gpu_step": B, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6)},
        }
        if want_roof:
            from a2f_amd import functional as a2f_fn
            per = {0: 1.0, 1: 3.0, 2: 6.0}
            mult = (per[a2f_fn.A2M_PREC_FWD] + per[a2f_fn.A2M_PREC_WGRAD] + per[a2f_fn.A2M_PREC_DGRAD]) / 3.0   # forward, weight- and data-gradient products: a third of the flops each
            alg = fps / world * A2M_MFLOP_PER_FRAME * 1e6 / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": round(mult * alg, 2), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(mult * alg / MFMA_BF16_PEAK_TF, 4),
                               "traffic": None, "algorithmic": round(alg, 2), "algorithmic_frac": round(alg / MFMA_BF16_PEAK_TF, 4), "executed_per_algorithmic": mult,
                               "note": "whole step: algorithmic fp32 flops (393 MFLOP/frame) / step time; executed = bf16 MFMA flops of the split products (forward x6, "
                                       "gradients x3) against the dense bf16 peak (2.5 PFLOP/s)"}
        if want_cpu:
            res["cpu_baseline"] = cpu_baseline_a2m()
        del step, model
        return res

    def run_faceformer_long(steps, warmup):
        """BASELINE configs[4]: 3000-frame sequences, per-vertex-L2 tolerance 1e-3.  The timed setting is the one that MEETS that tolerance
        (bf16x3 split products in GEMMs and attention: <= 1e-4 vs the oracle at T = 3000, tests/test_gpu_models.py::
        test_faceformer_config5_T3000_*).  The fp8 (e4m3) attention-forward modes are carried as labelled `variants`, each with its error
        measured here (eval-mode vertex offsets of one sequence against the bf16x3 path on the same weights and input)."""
        S, T = args.ff_long_seqs, args.ff_long_frames
        res, model, (audio, one_hot, templ) = run_faceformer(min(steps, 3), 2, S=S, T=T, config_name="BASELINE configs[4] (long context; per GPU)", cpu=False,
                                                             batch_points=(), long=True)
        res["parity_setting"] = {"attention": "bf16x3 split MFMA (forward and backward) - THE attention setting of this configuration: an fp8 form that meets 1e-3 "
                                              "needs three e4m3 terms per operand = six term products per k-step, the MFMA work of bf16x3 at best (block-scaled K = 64 "
                                              "instruction: 2x rate) plus conversions; measured +3 % per step in round 3 and removed", "tolerance": 1e-3,
                                 "pinned_by": "tests/test_gpu_models.py::test_faceformer_config5_T3000_encoder_vs_oracle (<= 1e-4 vs the CPU oracle)"}
        variants = []
        was_training = model.training
        specs = [(name, ap, label) for name, ap, label in
                 (("fp8_attention_forward", 2, "single e4m3 operands on v_mfma_f32_32x32x16_fp8_fp8"),)
                 if ap in getattr(be, "attn_precisions", (0, 1, 2))]
        # errors first, on the weights the reference output was computed with (the timed steps below update them)
        model.eval()
        errs = {}
        with torch.no_grad():
            ref = model(audio[:1], one_hot[:1], templ[:1]).view(T, -1) - templ[:1].reshape(1, -1)
            for name, ap, _label in specs:
                old_attn, be.attn_precision = be.attn_precision, ap
                try:
                    p2 = model(audio[:1], one_hot[:1], templ[:1]).view(T, -1) - templ[:1].reshape(1, -1)
                finally:
                    be.attn_precision = old_attn
                errs[name] = float((p2 - ref).double().norm() / ref.double().norm())
        model.train(was_training)
        for name, ap, label in specs:
            old_attn, be.attn_precision = be.attn_precision, ap
            try:
                torch.manual_seed(4321 + 7919 * rank)
                step = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-4)
                audio_f, one_hot_f, gt_f, templ_f = make_ff_batch(S, T, rank, device)
                el, _lv, _ti = timed(lambda: step(audio_f, one_hot_f, gt_f, templ_f), min(steps, 3), 2, barrier, world, device, budget_s=2.0, max_blocks=1)
                variants.append({"variant": name, "attention_forward": label, "attention_backward": "bf16x3 split MFMA" if ap == 3 else "plain bf16 MFMA",
                                 "vertex_offset_rel_l2_vs_parity_setting": float(f"{errs[name]:.3e}"), "meets_1e-3": bool(errs[name] <= 1e-3),
                                 "ms_per_step": round(1e3 * el / min(steps, 3), 3), "value": round(world * S * T * min(steps, 3) / el, 1)})
                del step, audio_f, one_hot_f, gt_f, templ_f
            finally:
                be.attn_precision = old_attn
        res["variants"] = variants
        del model
        return res

    t_start = time.perf_counter()

    def stage(msg):
        """Progress on stderr (rank 0): the driver's log shows where a long run is; stdout stays ONE JSON line."""
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f} s] {msg}", file=sys.stderr, flush=True)

    runners = {"voca": run_voca, "faceformer": run_faceformer_entry, "audio2mesh": run_audio2mesh, "faceformer_long": run_faceformer_long}
    stage(f"{args.workload}: start (world {world})")
    main_res = runners[args.workload](args.steps, args.warmup)
    stage(f"{args.workload}: done")
    result = {"metric": "mesh-frames/sec training (VOCASET 5023-vert)", "value": main_res.pop("value"), "unit": main_res.pop("unit"),
              "n_gpus": world, "steps": main_res.pop("steps"), "warmup": args.warmup, "ms_per_step": main_res.pop("ms_per_step"),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": main_res.pop("dtype"), "data": "synthetic"}
    # the north-star workloads' headline numbers as top-level keys right behind the standard ones (filled in below from `secondary`): BASELINE.json's target
    # sentence is on FaceFormer, so its step time / rate / GEMM-family share of the bf16 pipe must not hide at the end of a long line
    result["north_star"] = None
    result.update(main_res)
    if dist_info is not None:
        result["distributed"] = dist_info
    if not args.no_secondary:
        sec = []
        for name in ("faceformer", "audio2mesh", "faceformer_long", "voca"):
            if name == args.workload:
                continue
            if name == "voca":
                continue                                           # VOCA is the default headline; as a secondary it adds nothing new
            torch.cuda.empty_cache()
            # >= 3 timed steps per block unless the caller asks for fewer (the two-rank rehearsal test: over gloo a 385-MB gradient exchange takes seconds)
            r = runners[name](max(3 if args.secondary_steps >= 3 else 1, min(args.secondary_steps, args.steps)), min(3, args.warmup))
            r["model"] = name
            stage(f"secondary {name}: done")
            sec.append(r)
        result["secondary"] = sec
    ns = {}
    for r in [dict(result, model=args.workload)] + list(result.get("secondary") or []):
        roof = r.get("roofline") or {}
        e = {"ms_per_step": r.get("ms_per_step"), "frames_per_s": r.get("value"), "frames_per_gpu_step": (r.get("config") or {}).get("frames_per_gpu_step"),
             # VOCA's roofline.frac is its dominant KERNEL's fraction; the whole step's is roofline.whole_step_frac.  The MFMA-bound entries' roofline.frac
             # already is the whole step (executed flops); their algorithmic fraction rides along.
             "whole_step_frac": roof.get("whole_step_frac", roof.get("frac")), "bound": roof.get("bound")}
        if "whole_step_frac" in roof:
            e["dominant_kernel_frac"] = roof.get("frac")
        if "algorithmic_frac" in roof:
            e["algorithmic_frac"] = roof["algorithmic_frac"]
        if "gemm_family" in roof:
            e["gemm_family_frac"] = roof["gemm_family"].get("frac")
        pts = {p["seqs_per_step"]: p["ms_per_step"] for p in r.get("sequences_per_step_points") or []}
        if 1 in pts:
            e["batch1_ms_per_step"] = pts[1]                       # the reference's own batch = 1 step (train.py:33)
        ns[r["model"]] = e
    result["north_star"] = ns
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
