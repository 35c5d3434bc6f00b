#!/usr/bin/env python3
"""bench.py - mesh-frames/s of full training steps (fwd + loss + bwd + [all-reduce] + Adam) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` ; for N>1 launched by torch.distributed.run, one
rank per GPU (RCCL).  Prints ONE JSON line on rank 0.

Workload at N=1 = BASELINE.json configs[1]: VOCA (DeepSpeech-29 features, 8-subject one-hot), synthetic VOCASET batch
32 sequences x 600 frames = 19 200 frame windows (29,16) per step, 5023-vertex meshes, weak scaling (every rank runs
the same per-GPU batch on its own shard, seed 1234+rank; the only data-path exchange is the gradient all-reduce).
Other workloads (``--workload faceformer|audio2mesh``) are parity-test configurations, not bench lines.

Extra objects:
  roofline     - the dominant HIP kernel's ALGORITHMIC bytes per launch / its average launch duration, timed live with
                 HIP events recorded on the launch stream (ops.HipBackend.prof) over the timed steps of a separate,
                 identically shaped pass (so event overhead never sits inside the headline timing).
  cpu_baseline - the CPU oracle (oracle/, "port") timed on this node's host cores on a bounded sample of the same
                 workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

NV3 = 15069
FRAME_BYTES = NV3 * 4             # one fp32 mesh = 60 276 B

# algorithmic HBM bytes per frame of each tagged kernel (DESIGN.md "Kernels and rooflines")
ALGO_BYTES_PER_FRAME = {
    "head_fused": 3 * FRAME_BYTES + 50 * 4 * 2,      # template + gt read, pred written, H read, dH written
    "head_fwd": 2 * FRAME_BYTES + 50 * 4,            # template read, pred written
    "loss_fwd": 2 * FRAME_BYTES,                     # pred + gt read
    "loss_bwd": 3 * FRAME_BYTES,                     # pred + gt read, grad written
    "head_dH": FRAME_BYTES + 50 * 4,                 # grad read
    "head_dW": FRAME_BYTES + 50 * 4,                 # grad read
}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def make_voca_batch(n_seq: int, n_frames: int, rank: int, device, features: bool = False):
    """C2 of SURVEY §8(d): features (n_seq, n_frames, 29) ~ N(0,1) -> 16-frame windows (zero-padded ends), one subject
    per sequence, per-subject template ~ N(0,5), gt = template + N(0,0.5) (already x100-scaled units)."""
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


def cpu_baseline_voca(seconds: float = 12.0, B: int = 2048):
    """Oracle VOCA train step (fwd + VocaLoss + bwd + torch Adam) on the host cores, bounded sample."""
    from oracle import losses as o_loss, voca as o_voca, weights as W
    g = torch.Generator().manual_seed(99)
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
    # pick the intra-op thread count that is fastest on this host (oversubscribing a 256-thread host is 50x slower)
    ncpu = os.cpu_count() or 1
    best, best_t = None, float("inf")
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or n < 3:
        step(); n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 1), "unit": "mesh-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle VOCA train step (fwd+VocaLoss+bwd+Adam), B={B} frames x {n} steps in {dt:.1f}s, fp32 torch CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="voca", choices=["voca"])
    ap.add_argument("--seqs", type=int, default=32)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--input", default="windows", choices=["windows", "features"],
                    help="windows: x = (B,29,16) as the reference's Voca.forward takes it (default, the headline); features: x = (S,T,29), "
                         "the 16-frame windows are built inside the trunk kernel (SURVEY §8(f) N4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the ~90 kernels of a step eagerly instead of replaying a hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import a2f_amd
    from a2f_amd import ops
    from a2f_amd.trainer import TrainStep

    torch.manual_seed(1234)                                  # identical initial weights on every rank
    model = a2f_amd.Voca(n_verts=NV3, n_onehot=12).to(device)
    step = TrainStep(model, a2f_amd.VocaLoss(), lr=1e-4, use_graph=not args.no_graph)
    x, one_hot, gt, templ = make_voca_batch(args.seqs, args.frames, rank, device, features=args.input == "features")
    B = one_hot.shape[0]

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):                      # >= 3: two eager warm-up steps + the graph capture step
        out = step(x, one_hot, gt, templ)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(x, one_hot, gt, templ)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    loss_val = float(out["loss"])
    assert loss_val == loss_val, "loss is NaN"

    result = {
        "metric": "mesh-frames/sec training (VOCASET 5023-vert)",
        "value": round(world * B * args.steps / elapsed, 1),
        "unit": "mesh-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (head matrix products: bf16x3 split operands, f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"VOCA train step (fwd+VocaLoss+bwd+Adam), {args.seqs} seq x {args.frames} frames = {B} windows "
                               f"(29,16) per GPU per step{' built on chip from (S,T,29) features' if args.input == 'features' else ''}, 5023-vertex meshes, BASELINE configs[1]",
                   "frames_per_gpu_step": B, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6),
                   "launch": ("eager" if not step.use_graph else "hipGraph replay (fwd+loss+bwd) + eager all-reduce/Adam")
                             + ("; head-gradient all-reduce overlapped with the trunk backward" if step.overlap else "")},
    }

    if rank == 0 and not args.no_roofline:
        be = ops.backend()
        be.profile_head(True)                                  # HIP events on the launch stream around head_fused_kernel alone
        nprof = max(3, min(args.steps, 60))                    # the in-library ring holds 64 event pairs
        eager = TrainStep.__new__(TrainStep)                   # same model/optimizer, eager launches (events are not graph nodes)
        eager.__dict__.update(step.__dict__)
        eager.use_graph = False
        for _ in range(nprof):
            eager._fwd_bwd(x, one_hot, gt, templ)              # forward + loss + backward only: rank 0 runs this leg alone, so no collective

        torch.cuda.synchronize()
        head_ms, nl = be.profile_head_ms()
        be.profile_head(False)
        dom = "head_x3_kernel" if os.environ.get("A2F_HEAD_CFG", "5") == "5" else "head_fused_kernel"   # library default: cfg 5
        per = {dom: head_ms}
        bytes_per_launch = ALGO_BYTES_PER_FRAME["head_fused"] * B
        achieved = bytes_per_launch / (per[dom] * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            traffic = json.load(open(tj)).get(dom)
        result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                              "algorithmic_bytes_per_launch": bytes_per_launch,
                              "avg_launch_ms": round(per[dom], 4),
                              "launches_timed": nl}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_voca()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
