"""Minimal Lightning-free training step for the hot path (reference: src/model/lightning_model.py:145-161,209-213
and train.py:46-60).  One process per GPU; data parallel = one RCCL all-reduce of a flat fp32 gradient buffer per
step (xGMI), then one fused Adam launch over the flat parameter buffer.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist

from . import ops


class FlatAdam:
    """Adam(lr, weight_decay) over ONE flat fp32 buffer that aliases every parameter (and one that aliases every grad).

    Semantics = torch.optim.Adam as the reference configures it (lightning_model.py:209-213: weight_decay = lr/10,
    L2-style, default betas/eps).  ``params`` keep their identity: ``p.data`` becomes a view into the flat buffer,
    ``p.grad`` a view into the flat gradient buffer, so autograd accumulates straight into it and data-parallel
    training needs exactly one collective per step.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999),
                 eps: float = 1e-8, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.pg = process_group
        # rehearsal hook: run the collectives even in a one-rank group (exercises RCCL + hipGraph interplay on one GPU)
        self.force_collectives = os.environ.get("A2F_DP_FORCE_COLLECTIVES") == "1"
        # TEST HOOK (tests/test_gpu_models.py): `collective(tensor, async_op) -> work | None` in place of dist.all_reduce - a one-GPU box cannot hold two RCCL
        # ranks and a one-rank RCCL all-reduce launches no kernel, so the rehearsal of "a collective's kernels beside the backward" injects its own stream + kernel
        self.collective = None
        self.step_count = 0
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4           # 16-byte aligned segments
        self.numel = total
        self._offsets = {id(p): (o, (p.numel() + 3) // 4 * 4) for p, o in zip(self.params, offs)}
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view_as(p.data)
            p.grad = self.flat_g[o:o + n].view_as(p.data)
        self._grad_views = [p.grad for p in self.params]

    def zero_grad(self):
        self.flat_g.zero_()

    def release_grads(self):
        """Instead of zero_grad() in front of an autograd backward: with ``p.grad is None`` autograd hands every parameter its gradient
        tensor as it is - no fill of the flat buffer and no accumulation kernel per parameter (46 of them, ~5 us each, in the Audio2Mesh
        step; between the two backward passes of a pipelined FaceFormer step see ``stash_grads()``).  ``collect_grads()`` must follow the backward."""
        for p in self.params:
            p.grad = None

    def stash_grads(self):
        """Between two backward passes of one step taken with released gradients (the pipelined FaceFormer step, in front of its LAST encoder backward): set what
        has arrived aside and release again, so that the last pass hands its gradients out as they are too - the two are then added by ONE multi-tensor
        launch in ``collect_grads()`` instead of one in-place add per parameter inside the backward (212 launches of ~6 us per step, on the main stream's
        chain; same operands, same sum)."""
        stash = self._stash = getattr(self, "_stash", None) or {}
        for p, v in zip(self.params, self._grad_views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                prev = stash.get(id(p))
                stash[id(p)] = g if prev is None else prev.add_(g)
                p.grad = None

    def collect_grads(self):
        """Move the gradients autograd produced since ``release_grads()`` into the flat buffer (one multi-tensor copy; where ``stash_grads()`` set an earlier
        pass aside, copy + one multi-tensor add), zero the segments of parameters that received none (LayerDrop), and make ``p.grad`` the flat views again."""
        stash = getattr(self, "_stash", None) or {}
        dst, src, sum_dst, sum_b, missing = [], [], [], [], []
        for p, v in zip(self.params, self._grad_views):
            g, a = p.grad, stash.pop(id(p), None)
            if g is not None and g.data_ptr() == v.data_ptr():
                g = None if a is not None else g                   # (never both: a stashed pass leaves p.grad released)
            if a is not None:
                dst.append(v)
                src.append(a)
                if g is not None:
                    sum_dst.append(v)
                    sum_b.append(g)
            elif g is None:
                missing.append(v)
            elif g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
            if sum_b:
                torch._foreach_add_(sum_dst, sum_b)
            if src[0].is_cuda:
                # a stolen gradient may have been allocated and accumulated on the pipelined step's side streams; the copy above reads it on the
                # current stream and the tensor dies right here - tell the allocator, so the block is not handed out on its home stream while the copy runs
                cur = torch.cuda.current_stream(src[0].device)
                for g in src + sum_b:
                    g.record_stream(cur)
        if missing:
            torch._foreach_zero_(missing)

    def world(self) -> int:
        return dist.get_world_size(self.pg) if dist.is_available() and dist.is_initialized() else 1

    def allreduce_grads(self, lo: int = 0, hi: Optional[int] = None, async_op: bool = False):
        """Sum the flat gradient range [lo, hi) over the ranks (the 1/world factor lives in the Adam kernel)."""
        if self.collective is not None:
            return self.collective(self.flat_g[lo:self.numel if hi is None else hi], async_op)
        if self.world() > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized()):
            return dist.all_reduce(self.flat_g[lo:self.numel if hi is None else hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        return None

    def tail_segment(self, params) -> Optional[int]:
        """Offset at which ``params`` start if they occupy exactly the tail of the flat buffer, else None."""
        spans = sorted(self._offsets[id(p)] for p in params if id(p) in self._offsets)
        if len(spans) != len(list(params)) or not spans:
            return None
        pos = spans[0][0]
        for o, n in spans:
            if o != pos:
                return None
            pos += n
        return spans[0][0] if pos == self.numel else None

    def step(self, reduced: bool = False):
        """All-reduce (if distributed and not already ``reduced`` by the caller) + fused Adam.  The 1/world averaging is
        folded into the Adam kernel."""
        if not reduced:
            self.allreduce_grads()
        self.step_count += 1
        from . import encoder_ops
        encoder_ops.bump_weights_epoch()                # the kernel below rewrites every parameter through its raw pointer: cached weight images are stale
        ops.backend().adam_step(self.flat_p, self.flat_g, self.m, self.v, self.numel, self.lr, self.betas[0], self.betas[1],
                                self.eps, self.wd, 1.0 / self.world(), self.step_count)


class BucketedAllReduce:
    """Gradient exchange overlapped with the backward (what the reference gets from Lightning's implicit DDP, train.py:48-59): the flat
    gradient buffer is cut into contiguous buckets in parameter order - for FaceFormer: encoder front end | one per Transformer layer |
    decoder side - and each bucket's sum all-reduce is started asynchronously as soon as the backward has moved past it.

    Order.  The backward of a sequential network finishes parameters in REVERSE parameter order, so when a gradient of bucket b arrives
    every bucket above b is complete (its AccumulateGrad nodes run before the next layer's backward node: leaf accumulations have the
    highest priority in the autograd engine).  Buckets are therefore launched strictly in descending index order - on every rank the same
    sequence of collectives whatever LayerDrop skipped on that rank (a skipped layer's gradients simply stay zero) - and ``finish()``
    launches what is left (the first bucket) and waits for all of them.  The launches come from autograd's post-accumulate hooks; the
    collective library orders each all-reduce behind the kernels already queued on the calling thread's current stream."""

    def __init__(self, opt: FlatAdam, named_params, key_fn):
        self.opt = opt
        self.bounds, self.bucket_of = [], {}
        last_key, lo = object(), 0
        for name, p in named_params:
            if id(p) not in opt._offsets:
                continue
            off, n = opt._offsets[id(p)]
            k = key_fn(name)
            if k != last_key:
                if self.bounds:
                    self.bounds[-1] = (self.bounds[-1][0], off)
                self.bounds.append((off, off))
                last_key = k
            self.bucket_of[id(p)] = len(self.bounds) - 1
            self.bounds[-1] = (self.bounds[-1][0], off + n)
        assert self.bounds and self.bounds[0][0] == 0 and self.bounds[-1][1] == opt.numel and all(a[1] == b[0] for a, b in zip(self.bounds, self.bounds[1:])), \
            "buckets must tile the flat buffer in parameter order"
        self.active = False
        self.next = len(self.bounds) - 1
        self.works = []
        self.launch_log = []                                   # bucket indices in launch order (tests)
        # the hooks hold this object (and through it the FlatAdam and its four flat buffers): keep the handles so that close() can take them
        # off the parameters when the step that owns this exchange goes away (bench.py builds several TrainSteps on one model)
        self._handles = []
        for name, p in named_params:
            if id(p) in self.bucket_of:
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def close(self):
        """Remove the gradient hooks (idempotent)."""
        for h in self._handles:
            h.remove()
        self._handles = []
        self.active = False

    def begin(self):
        """Call right before the LAST backward pass of the step (gradients of earlier passes are already accumulated)."""
        self.active, self.next, self.works, self.launch_log = True, len(self.bounds) - 1, [], []

    def _launch_down_to(self, b):
        while self.next > b:
            lo, hi = self.bounds[self.next]
            w = self.opt.allreduce_grads(lo, hi, async_op=True)
            if w is not None:
                self.works.append(w)
            self.launch_log.append(self.next)
            self.next -= 1

    def _hook(self, p):
        if self.active:
            b = self.bucket_of[id(p)]
            if b > self.next:
                # the order assumption above does not hold for this model (a parameter registered early but used late, e.g. a final norm filed
                # under the front bucket): bucket b went out before this gradient existed.  Nothing can repair a sum that is already on the wire.
                raise RuntimeError(f"BucketedAllReduce: a gradient of bucket {b} arrived after that bucket's all-reduce was started (buckets are "
                                   f"launched in reverse parameter order); this model needs A2F_DP_BUCKETS=0 (one all-reduce after the backward)")
            self._launch_down_to(b)

    def finish(self) -> bool:
        """Launch the remaining buckets, wait for every all-reduce.  Returns True when the gradients are reduced (begin() was called)."""
        if not self.active:
            return False
        self._launch_down_to(-1)
        for w in self.works:
            w.wait()
        self.active, self.works = False, []
        return True


_CAPTURE_STREAMS = {}


class StepOutput(dict):
    """{"loss", "rec_loss", "vel_loss"} of a step; "err" (mse_error of lightning_model.py:119-125 == rec_loss / 3 for
    3-vectors) is derived on every access instead of costing an extra launch in every step.  It is NOT cached: under
    hipGraph replay the same StepOutput object wraps static loss buffers that every replay rewrites."""

    def __missing__(self, key):
        if key != "err":
            raise KeyError(key)
        return self["rec_loss"].detach() / 3.0


class TrainStep:
    """fwd -> loss -> bwd -> (all-reduce) -> Adam, mirroring Audio2FaceModel.training_step + configure_optimizers."""

    def __init__(self, model: torch.nn.Module, loss_fn, lr: float, process_group=None, fused: bool = True,
                 use_graph: bool = False, overlap: Optional[bool] = None):
        self.model = model
        self.loss_fn = loss_fn
        # use_graph: capture zero_grad + forward + loss + backward (~90 launches) into one hipGraph after a warm-up and
        # replay it every step; the all-reduce and the Adam launch stay eager.  The caller must then pass the SAME
        # input tensors every step (copy each new batch into them) - the usual static-buffer contract of graph replay.
        self.use_graph = use_graph
        self._graph = None
        self._graph_inputs = None
        self._graph_out = None
        self._warm = 0
        self._side = None
        # fused = head + loss + head-backward in one HIP pass when the model offers it and the loss is the stock VocaLoss
        self.fused = fused and hasattr(model, "forward_with_loss") and type(loss_fn).__name__ in ("VocaLoss", "FaceFormerLoss")
        self.opt = FlatAdam(model.parameters(), lr=lr, weight_decay=lr / 10, process_group=process_group)
        # overlap: split the direct step after the fused head, start the all-reduce of the head's gradient segment (the
        # tail of the flat buffer, 92 % of the bytes for VOCA) and run the trunk backward underneath it; the small trunk
        # segment is reduced afterwards.  Default: on when distributed (A2F_DP_OVERLAP=0 disables), off otherwise.
        self._head_off = None
        if self.fused and hasattr(model, "train_step_direct") and hasattr(model, "early_grad_params"):
            self._head_off = self.opt.tail_segment(model.early_grad_params())
        if overlap is None:
            overlap = (self.opt.world() > 1 or self.opt.force_collectives) and os.environ.get("A2F_DP_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and self._head_off is not None
        # FaceFormer with several sequences per step: two half-batches on two streams (Faceformer.train_step_pipelined); A2F_FF_PIPELINE=0
        # keeps the single-stream order
        # (measured on MI355X with the round-1 recurrence, 17 ms per half at T = 600: 64 sequences 179.5 -> 171.5 ms.  With the 8-wave forward and the
        # blocked record-form backward the recurrence is 11 ms at T = 600 and the halves' doubled per-launch tails cost as much as hiding it gains:
        # 64 x 600 155.9 ms single-stream vs 158-161 ms pipelined inside the full bench run; at 3000 frames (89 ms per half) it pays: 12 x 3000
        # 251 -> 238 ms.  Round 4: with the encoder's products on image operands the step is 30 ms shorter, the recurrence is 6 % of it again and hiding
        # it pays at 600 frames too: 64 x 600 131.0 -> 125.8 ms, same box.)  So: from 600 frames per sequence and 48 x 600 frames per step up.
        self.pipeline = os.environ.get("A2F_FF_PIPELINE", "1") != "0"
        self.pipeline_min_seqs = int(os.environ.get("A2F_FF_PIPELINE_MIN_SEQS", "48"))    # in sequences of 600 frames: the test is on frames per step
        self.pipeline_min_frames = int(os.environ.get("A2F_FF_PIPELINE_MIN_FRAMES", "600"))
        self._graph2 = None
        self._rand = None             # dropout.StepRandomness of a captured FaceFormer step (_static_randomness)
        # models that expose `grad_bucket_key(name)` (FaceFormer: 386 MB of gradients) exchange them bucket by bucket under the backward
        # (A2F_DP_BUCKETS=0: one blocking all-reduce after the backward, as round 2 did)
        self.buckets = None
        self._mid_collect = False
        if (hasattr(model, "grad_bucket_key") and (self.opt.world() > 1 or self.opt.force_collectives)
                and os.environ.get("A2F_DP_BUCKETS", "1") != "0"):
            self.buckets = BucketedAllReduce(self.opt, list(model.named_parameters()), model.grad_bucket_key)
        if self.buckets is not None and use_graph:
            # autograd hooks that launch collectives cannot live inside a hipGraph capture (and a replay would not run them): a captured step
            # exchanges its gradients with the one blocking all-reduce inside FlatAdam.step() instead
            self.buckets.close()
            self.buckets = None

    def close(self):
        """Detach this step from the model (gradient hooks of the bucketed exchange); the model and its parameters stay usable."""
        if getattr(self, "buckets", None) is not None:
            self.buckets.close()
            self.buckets = None
        self._graph = self._graph2 = None     # a captured graph reads the device buffers of self._rand: they go together
        self._rand = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _direct(self, ins, phase):
        lf = getattr(self.loss_fn, "loss", self.loss_fn)
        x, one_hot, gt, template = ins
        _pred, out3 = self.model.train_step_direct(x, one_hot, template, gt, lf.k_rec, lf.k_vel, phase=phase)
        return StepOutput({"loss": out3[0], "rec_loss": out3[1], "vel_loss": out3[2]})

    def _finish_overlapped(self, second_phase):
        """head segment all-reduce (async) || trunk backward, then the trunk segment, then Adam."""
        work = self.opt.allreduce_grads(self._head_off, None, async_op=True)
        second_phase()
        self.opt.allreduce_grads(0, self._head_off)
        if work is not None:
            work.wait()
        self.opt.step(reduced=True)

    @staticmethod
    def unpack_batch(batch: Dict[str, torch.Tensor], scale: float = 100.0):
        """lightning_model.py:145-148 (returns new tensors; the caller's dict entries are rebound, not mutated)."""
        batch["verts"] = batch["verts"] * scale
        batch["template_vert"] = batch["template_vert"] * scale
        return batch["audio"], batch["one_hot"], batch["verts"], batch["template_vert"]

    def __call__(self, x, one_hot, gt, template) -> Dict[str, torch.Tensor]:
        if self.use_graph and x.is_cuda:
            return self._graph_step(x, one_hot, gt, template)
        if self.overlap:
            ins = (x, one_hot, gt, template)
            loss = self._direct(ins, 1)
            self._finish_overlapped(lambda: self._direct(ins, 2))
            return loss
        loss = self._fwd_bwd(x, one_hot, gt, template)
        self.opt.step(reduced=self.buckets.finish() if self.buckets is not None else False)
        return loss

    def _static_randomness(self, x):
        """Captured FaceFormer step: the step's random state (mask seeds, LayerDrop decisions, SpecAugment rows) must live on the device - a value drawn on
        the host inside the captured region would be frozen into the graph.  The dropout.StepRandomness belongs to THIS step object (`self._rand`): created once,
        refreshed (host draws + non-blocking copies on the current stream, OUTSIDE the graph) before every warm-up / capture / replay, and visible to the model
        only while this step runs its Python forward (`_installed_randomness`): a later eager forward of the same model draws on the host as ever."""
        enc = getattr(self.model, "audio_encoder", None)
        if enc is None or not hasattr(enc, "static_rand") or not x.is_cuda or x.dim() != 2:
            return None
        from .dropout import StepRandomness
        S, T = x.shape[0], x.shape[1] * 60 // 16000
        sr = self._rand
        if sr is None or (sr.S, sr.T) != (S, T):
            sr = self._rand = StepRandomness(x.device, len(enc.encoder.layers), S, T)
        sr.refresh(enc.config, train_encoder=bool(enc.training and enc.train_regularisers and enc.injected_masks is None),
                   train_decoder=bool(self.model.training and getattr(self.model, "decoder_regularisers", False)))
        return sr

    def _installed_randomness(self, sr):
        """Context: the model's encoder sees `sr` as its device-resident random state for the duration of a warm-up step or a capture, and what it had before
        (normally None) afterwards - also when the body raises."""
        import contextlib
        enc = getattr(self.model, "audio_encoder", None)
        if sr is None or enc is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def scope():
            old, enc.static_rand = enc.static_rand, sr
            try:
                yield
            finally:
                enc.static_rand = old
        return scope()

    def _graph_step(self, x, one_hot, gt, template):
        ins = (x, one_hot, gt, template)
        sr = self._static_randomness(x)
        if self._graph is None:
            # Warm-up and capture run on ONE side stream: autograd's AccumulateGrad nodes remember the stream their
            # gradient buffers were last used on, and a capture that has to synchronise with another stream is illegal.
            if self._side is None:
                # ONE capture stream per device for every TrainStep of the process: HIP maps streams onto a handful of hardware queues, and each
                # further stream can land on the queue another one already uses (bench.py builds several steps in one process: with a stream per
                # step the pipelined FaceFormer step behind them ran 153 ms instead of 125)
                dev = torch.cuda.current_device()
                if dev not in _CAPTURE_STREAMS:
                    _CAPTURE_STREAMS[dev] = torch.cuda.Stream()
                self._side = _CAPTURE_STREAMS[dev]
                self._side.wait_stream(torch.cuda.current_stream())
            if self._warm < 2:                                  # eager warm-up: allocator pools, LDS attributes, autotune-free
                self._warm += 1
                self._side.wait_stream(torch.cuda.current_stream())     # the refresh above copied this step's random state on the current stream
                with torch.cuda.stream(self._side), self._installed_randomness(sr):
                    loss = self._direct(ins, 1) if self.overlap else self._fwd_bwd(*ins)
                torch.cuda.current_stream().wait_stream(self._side)
                if self.overlap:
                    def second():
                        self._side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(self._side):
                            self._direct(ins, 2)
                        torch.cuda.current_stream().wait_stream(self._side)
                    self._finish_overlapped(second)
                else:
                    self.opt.step()
                self._side.wait_stream(torch.cuda.current_stream())
                return loss
            torch.cuda.synchronize()
            # thread_local: the capture must not be invalidated by HIP calls of other threads of this process (RCCL's
            # watchdog polls events while the ranks are between collectives)
            try:
                g = torch.cuda.CUDAGraph()
                with self._installed_randomness(sr), torch.cuda.graph(g, stream=self._side, capture_error_mode="thread_local"):
                    out = self._direct(ins, 1) if self.overlap else self._fwd_bwd(*ins)
                if self.overlap:                                # second graph: trunk backward, same memory pool
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, stream=self._side, pool=g.pool(), capture_error_mode="thread_local"):
                        self._direct(ins, 2)
                    self._graph2 = g2
            except RuntimeError as e:
                # a failed capture executes nothing: keep training with eager launches of the same HIP kernels (slower
                # launches, same arithmetic) instead of aborting the job
                import warnings
                warnings.warn(f"hipGraph capture of the train step failed ({e}); continuing with eager launches")
                torch.cuda.synchronize()
                self.use_graph = False
                self._graph = self._graph2 = None
                self._rand = None                                     # the eager step draws on the host again
                return self.__call__(*ins)
            self._graph, self._graph_inputs, self._graph_out = g, ins, out
        else:
            if any(a.data_ptr() != b.data_ptr() for a, b in zip(ins, self._graph_inputs)):
                raise RuntimeError("TrainStep(use_graph=True) replays a captured graph: pass the same input tensors each step")
        self._graph.replay()
        if self.overlap:
            self._finish_overlapped(self._graph2.replay)
        else:
            self.opt.step()
        return self._graph_out

    def _fwd_bwd(self, x, one_hot, gt, template) -> Dict[str, torch.Tensor]:
        if self.fused and hasattr(self.model, "train_step_direct"):
            # autograd-free launch sequence that overwrites the flat gradient views (no zero fill, no accumulation kernels)
            return self._direct((x, one_hot, gt, template), 0)
        # bucketed exchange: its hooks all-reduce ranges of the flat buffer while the LAST backward pass runs, so that pass must accumulate there
        # directly.  A pipelined step has an earlier pass: its gradients are still taken as autograd hands them out and moved into the flat buffer
        # right before the last pass (_begin_exchange) - no zero fill and no accumulation kernel per parameter for the first half (215 launches)
        self._mid_collect = self.buckets is not None and self._pipelined(x)
        stolen = self.buckets is None or self._mid_collect
        self._mid_collect = self._mid_collect and stolen
        if stolen:
            self.opt.release_grads()
        else:
            self.opt.zero_grad()
        try:
            return self._fwd_bwd_autograd(x, one_hot, gt, template)
        finally:
            if stolen:
                self.opt.collect_grads()

    def _pipelined(self, x) -> bool:
        """FaceFormer with several sequences per step on a GPU: the step runs as two half-batches on two streams (Faceformer.train_step_pipelined)."""
        if not (self.fused and hasattr(self.model, "forward_with_loss_batch") and hasattr(self.model, "train_step_pipelined")):
            return False
        if not (self.pipeline and x.is_cuda and x.dim() == 2 and x.shape[0] >= 2):
            return False
        if self.use_graph:                     # a captured step is one stream and one StepRandomness for the whole batch: the single-stream order
            return False
        frames = x.shape[1] * 60 // 16000
        return frames >= self.pipeline_min_frames and x.shape[0] * frames >= self.pipeline_min_seqs * 600

    def _begin_exchange(self):
        """Right before the last backward pass of a pipelined step under the bucketed exchange."""
        if self._mid_collect:
            self.opt.collect_grads()           # the first half's gradients -> flat buffer; p.grad = the flat views the hooks reduce
        self.buckets.begin()

    def _fwd_bwd_autograd(self, x, one_hot, gt, template) -> Dict[str, torch.Tensor]:
        if self.fused:
            lf = getattr(self.loss_fn, "loss", self.loss_fn)          # FaceFormerLoss wraps a VocaLoss
            if hasattr(self.model, "forward_with_loss_batch") and x.dim() == 2 and x.shape[0] > 1:
                # S independent batch-1 FaceFormer problems per step (train.py:33 fixes batch = 1)
                if self._pipelined(x):
                    # two half-batches on two streams: each half's decoder recurrence runs under the other half's encoder work
                    loss = self.model.train_step_pipelined(x, one_hot, template, gt, lf.k_rec, lf.k_vel,
                                                           before_last_backward=self._begin_exchange if self.buckets is not None else self.opt.stash_grads)
                    loss["err"] = loss["rec_loss"] / 3.0
                    return loss
                _pred, loss = self.model.forward_with_loss_batch(x, one_hot, template, gt, lf.k_rec, lf.k_vel)
            else:
                _pred, loss = self.model.forward_with_loss(x, one_hot, template, gt, lf.k_rec, lf.k_vel, unit_grad=True)
        else:
            pred = self.model(x, one_hot, template)
            loss = self.loss_fn(pred, gt)
        if self.buckets is not None:
            self.buckets.begin()
        loss["loss"].backward()
        # mse_error (lightning_model.py:119-125) == rec_loss / 3 for 3-vectors: no extra pass over pred/gt
        loss["err"] = loss["rec_loss"].detach() / 3.0
        return loss
