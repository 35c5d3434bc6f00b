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
def _pick_threads(step):
    """Fastest intra-op thread count on this host (oversubscribing a 256-thread host is 50x slower)."""
    import torch
    ncpu = os.cpu_count() or 1
    best, best_t = None, float("inf")
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    return best


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
def timed(step_fn, steps, warmup, barrier, world, device):
    import torch
    import torch.distributed as dist
    out = None
    for _ in range(warmup):
        out = step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step_fn()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    loss_val = float(out["loss"])
    assert loss_val == loss_val, "loss is NaN"
    return elapsed, loss_val


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
        elapsed, loss_val = timed(lambda: step(x, one_hot, gt, templ), steps, max(warmup, 3), barrier, world, device)   # >= 3: two eager warm-up steps + the capture step
        res = {
            "value": round(world * B * steps / elapsed, 1), "unit": "mesh-frames/s", "steps": steps,
            "ms_per_step": round(1e3 * elapsed / steps, 4),
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
            dom = "head_x3_kernel" if os.environ.get("A2F_HEAD_CFG", "5") == "5" else "head_fused_kernel"   # library default: cfg 5
            bytes_per_launch = ALGO_BYTES_PER_FRAME["head_fused"] * B
            achieved = bytes_per_launch / (head_ms * 1e-3) / 1e9
            traffic = None
            tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tj):
                traffic = json.load(open(tj)).get(dom)
            res["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(head_ms, 4), "launches_timed": nl,
                               "measured_stream_GBs": stream_rates,
                               "whole_step_frac": round(B * steps / elapsed * 182732 / 1e9 / HBM_PEAK_GBS, 4)}
        if want_cpu:
            res["cpu_baseline"] = cpu_baseline_voca()
        del step, model
        return res

    # ------------------------------------------------------------------ FaceFormer (BASELINE configs[2])
    def run_faceformer(steps, warmup, S=None, T=None, fp8_attention=False, config_name="BASELINE configs[2]", cpu=True):
        torch.manual_seed(1234)
        S, T = S or args.ff_seqs, T or args.ff_frames
        old_attn = be.attn_precision
        if fp8_attention:                                        # configs[4]: fp8 (e4m3) QK^T / PV products in the encoder's attention forward
            be.attn_precision = 2
        model = a2f_amd.Faceformer(n_verts=NV3, n_onehot=12).to(device)
        with torch.no_grad():                                    # the reference zero-initialises both vertice maps (faceformer.py:132-135):
            model.vertice_map_r.weight.normal_(0, 0.1)           # non-zero values so the recurrence and its backward do real work
            model.vertice_map.weight.normal_(0, 0.001)
        if args.encoder_precision == "bf16":
            model.encoder_autocast = torch.bfloat16
        model.train()                                            # Lightning's fit puts the reference's module in train mode
        torch.manual_seed(4321 + 7919 * rank)                    # same initial weights on every rank (seed above), different dropout masks per rank
        if args.ff_deterministic:                                # A/B: the deterministic (eval-mode) function under the same step
            model.audio_encoder.train_regularisers = model.decoder_regularisers = False
        step = TrainStep(model, a2f_amd.FaceFormerLoss(), lr=1e-4)
        audio, one_hot, gt, templ = make_ff_batch(S, T, rank, device)
        elapsed, loss_val = timed(lambda: step(audio, one_hot, gt, templ), steps, max(warmup, 2), barrier, world, device)
        fps = world * S * T * steps / elapsed
        enc_path = getattr(model.audio_encoder, "compute_path", "torch ops (rocBLAS / MIOpen / SDPA)")
        res = {
            "value": round(fps, 1), "unit": "mesh-frames/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3),
            "dtype": "f32 storage; " + ("encoder products as bf16 MFMA (16-mixed), fp32 accumulate / softmax / LayerNorm" if args.encoder_precision == "bf16"
                                        else getattr(model.audio_encoder, "dtype_note", "encoder in fp32 torch ops"))
                     + ("; encoder attention forward QK^T / PV as fp8 e4m3 MFMA (backward bf16)" if fp8_attention else ""),
            "config": {"workload": f"FaceFormer train step (wav2vec2-base encoder + incremental decoder + fused head/FaceFormerLoss, bwd, Adam over 96.4 M "
                                   f"parameters), {S} independent batch-1 sequences x {T} frames per GPU per step, {config_name}",
                       "frames_per_gpu_step": S * T, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6), "encoder": enc_path,
                       "train_mode": ("deterministic function (regularisers off, --ff-deterministic)" if args.ff_deterministic else
                                      "reference train mode: wav2vec2 dropout x6 kinds of sites / LayerDrop 0.1 / SpecAugment, decoder + PPE dropout 0.1 "
                                      "(one mask per frame row), masks drawn on the device every step")},
        }
        if want_roof:
            # whole step against the matrix peak (857 MFLOP/frame algorithmic) + the tagged GEMM launches alone
            ach = fps / world * FF_MFLOP_PER_FRAME * 1e6 / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
                    "traffic": None, "note": "whole step, algorithmic fp32 flops (857 MFLOP/frame) / step time, against the exact-fp32 matrix peak; "
                                             "split-bf16 kernels execute 3 bf16 MFMA flops per algorithmic flop: bf16_pipe_frac = 3 x achieved / 2500"}
            roof["bf16_pipe_frac"] = round(3 * ach / MFMA_BF16_PEAK_TF, 4)
            # per-launch HIP events are only meaningful when nothing else shares the GPU: this pass runs the same kernels with the
            # two-stream half-batch pipeline and the weight-gradient lane switched off (one stream, launches back to back)
            from a2f_amd import encoder_ops
            old_lane, old_pipe = encoder_ops.WGRAD_STREAM, step.pipeline
            encoder_ops.WGRAD_STREAM, step.pipeline = False, False
            be.prof, be.prof_flops = {}, {}
            for _ in range(2):
                step._fwd_bwd(audio, one_hot, gt, templ)         # forward + loss + backward only: rank 0 runs this leg alone, so no collective (no Adam either)
            torch.cuda.synchronize()
            encoder_ops.WGRAD_STREAM, step.pipeline = old_lane, old_pipe
            tf, nl, ms = gemm_prof_summary(be)
            if tf is not None:
                roof["kernel"] = {"name": "gemm_mfma_kernel (all encoder / decoder-side GEMM launches of 2 steps)", "launches_timed": nl,
                                  "total_ms": round(ms, 3), "achieved_TFLOPs": round(tf, 2), "frac_of_f32_matrix_peak": round(tf / MFMA_F32_PEAK_TF, 4),
                                  "bf16_pipe_frac": round(3 * tf / MFMA_BF16_PEAK_TF, 4)}
            hev = be.prof.get("head_fused", [])
            if hev:
                hms = sum(a.elapsed_time(b) for a, b in hev) / len(hev)
                hb = ALGO_BYTES_PER_FRAME["head_fused_ff"] * (2 * S * T / len(hev))          # frames per launch over the 2 profiled steps (single stream)
                roof["head_hbm"] = {"kernel": "head_x3_kernel<64> + fold", "achieved": round(hb / (hms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(hb / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(hms, 4), "launches_timed": len(hev)}
            be.prof = None
            res["roofline"] = roof
        if want_cpu and cpu:
            res["cpu_baseline"] = cpu_baseline_faceformer()
        be.attn_precision = old_attn
        del step, model
        return res

    # ------------------------------------------------------------------ Audio2Mesh
    def run_audio2mesh(steps, warmup):
        torch.manual_seed(1234)
        B = args.a2m_batch
        model = a2f_amd.Audio2Mesh(n_verts=NV3, n_onehot=12).to(device).train()
        step = TrainStep(model, a2f_amd.VocaLoss(), lr=1e-4, use_graph=not args.no_graph)
        x, one_hot, gt, templ = make_a2m_batch(B, rank, device)
        elapsed, loss_val = timed(lambda: step(x, one_hot, gt, templ), steps, max(warmup, 3), barrier, world, device)
        fps = world * B * steps / elapsed
        res = {
            "value": round(fps, 1), "unit": "mesh-frames/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3),
            "dtype": "f32 storage and accumulate; convolution products as split-bf16 MFMA (fp32-class)",
            "config": {"workload": f"Audio2Mesh train step (train-mode BatchNorm, fwd+VocaLoss+bwd+Adam), B = {B} windows (52,32) per GPU per step",
                       "frames_per_gpu_step": B, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6)},
        }
        if want_roof:
            ach = fps / world * A2M_MFLOP_PER_FRAME * 1e6 / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
                               "traffic": None, "note": "whole step, algorithmic fp32 flops (393 MFLOP/frame) / step time, against the exact-fp32 matrix peak"}
        if want_cpu:
            res["cpu_baseline"] = cpu_baseline_a2m()
        del step, model
        return res

    def run_faceformer_long(steps, warmup):                      # BASELINE configs[4]: 3000-frame sequences, fp8 MFMA attention
        return run_faceformer(min(steps, 3), 2, S=args.ff_long_seqs, T=args.ff_long_frames, fp8_attention=True,
                              config_name="BASELINE configs[4] (long context, fp8 attention; per GPU)", cpu=False)

    runners = {"voca": run_voca, "faceformer": run_faceformer, "audio2mesh": run_audio2mesh, "faceformer_long": run_faceformer_long}
    main_res = runners[args.workload](args.steps, args.warmup)
    result = {"metric": "mesh-frames/sec training (VOCASET 5023-vert)", "value": main_res.pop("value"), "unit": main_res.pop("unit"),
              "n_gpus": world, "steps": main_res.pop("steps"), "warmup": args.warmup, "ms_per_step": main_res.pop("ms_per_step"),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": main_res.pop("dtype"), "data": "synthetic"}
    result.update(main_res)
    if not args.no_secondary:
        sec = []
        for name in ("faceformer", "audio2mesh", "faceformer_long", "voca"):
            if name == args.workload:
                continue
            if name == "voca":
                continue                                           # VOCA is the default headline; as a secondary it adds nothing new
            torch.cuda.empty_cache()
            r = runners[name](max(3, min(args.secondary_steps, args.steps)), 3)
            r["model"] = name
            sec.append(r)
        result["secondary"] = sec
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
