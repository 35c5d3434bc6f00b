"""Minimal Lightning-free training step for the hot path (reference: src/model/lightning_model.py:145-161,209-213
and train.py:46-60).  One process per GPU; data parallel = one RCCL all-reduce of a flat fp32 gradient buffer per
step (xGMI), then one fused Adam launch over the flat parameter buffer.
"""
from __future__ import annotations

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
        self.step_count = 0
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4           # 16-byte aligned segments
        self.numel = total
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view_as(p.data)
            p.grad = self.flat_g[o:o + n].view_as(p.data)

    def zero_grad(self):
        self.flat_g.zero_()

    def world(self) -> int:
        return dist.get_world_size(self.pg) if dist.is_available() and dist.is_initialized() else 1

    def allreduce_grads(self):
        if self.world() > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)

    def step(self):
        """All-reduce (if distributed) + fused Adam.  The 1/world averaging is folded into the Adam kernel."""
        self.allreduce_grads()
        self.step_count += 1
        ops.backend().adam_step(self.flat_p, self.flat_g, self.m, self.v, self.numel, self.lr, self.betas[0], self.betas[1],
                                self.eps, self.wd, 1.0 / self.world(), self.step_count)


class StepOutput(dict):
    """{"loss", "rec_loss", "vel_loss"} of a step; "err" (mse_error of lightning_model.py:119-125 == rec_loss / 3 for
    3-vectors) is derived on first access instead of costing an extra launch in every step."""

    def __missing__(self, key):
        if key != "err":
            raise KeyError(key)
        v = self["rec_loss"].detach() / 3.0
        self[key] = v
        return v


class TrainStep:
    """fwd -> loss -> bwd -> (all-reduce) -> Adam, mirroring Audio2FaceModel.training_step + configure_optimizers."""

    def __init__(self, model: torch.nn.Module, loss_fn, lr: float, process_group=None, fused: bool = True,
                 use_graph: bool = False):
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

    @staticmethod
    def unpack_batch(batch: Dict[str, torch.Tensor], scale: float = 100.0):
        """lightning_model.py:145-148 (returns new tensors; the caller's dict entries are rebound, not mutated)."""
        batch["verts"] = batch["verts"] * scale
        batch["template_vert"] = batch["template_vert"] * scale
        return batch["audio"], batch["one_hot"], batch["verts"], batch["template_vert"]

    def __call__(self, x, one_hot, gt, template) -> Dict[str, torch.Tensor]:
        if self.use_graph and x.is_cuda:
            return self._graph_step(x, one_hot, gt, template)
        loss = self._fwd_bwd(x, one_hot, gt, template)
        self.opt.step()
        return loss

    def _graph_step(self, x, one_hot, gt, template):
        ins = (x, one_hot, gt, template)
        if self._graph is None:
            # Warm-up and capture run on ONE side stream: autograd's AccumulateGrad nodes remember the stream their
            # gradient buffers were last used on, and a capture that has to synchronise with another stream is illegal.
            if self._side is None:
                self._side = torch.cuda.Stream()
                self._side.wait_stream(torch.cuda.current_stream())
            if self._warm < 2:                                  # eager warm-up: allocator pools, LDS attributes, autotune-free
                self._warm += 1
                with torch.cuda.stream(self._side):
                    loss = self._fwd_bwd(*ins)
                torch.cuda.current_stream().wait_stream(self._side)
                self.opt.step()
                self._side.wait_stream(torch.cuda.current_stream())
                return loss
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._side):
                out = self._fwd_bwd(*ins)
            self._graph, self._graph_inputs, self._graph_out = g, ins, out
        else:
            if any(a.data_ptr() != b.data_ptr() for a, b in zip(ins, self._graph_inputs)):
                raise RuntimeError("TrainStep(use_graph=True) replays a captured graph: pass the same input tensors each step")
        self._graph.replay()
        self.opt.step()
        return self._graph_out

    def _fwd_bwd(self, x, one_hot, gt, template) -> Dict[str, torch.Tensor]:
        if self.fused and hasattr(self.model, "train_step_direct"):
            # autograd-free launch sequence that overwrites the flat gradient views (no zero fill, no accumulation kernels)
            lf = getattr(self.loss_fn, "loss", self.loss_fn)
            _pred, out3 = self.model.train_step_direct(x, one_hot, template, gt, lf.k_rec, lf.k_vel)
            return StepOutput({"loss": out3[0], "rec_loss": out3[1], "vel_loss": out3[2]})
        self.opt.zero_grad()
        if self.fused:
            lf = getattr(self.loss_fn, "loss", self.loss_fn)          # FaceFormerLoss wraps a VocaLoss
            _pred, loss = self.model.forward_with_loss(x, one_hot, template, gt, lf.k_rec, lf.k_vel, unit_grad=True)
        else:
            pred = self.model(x, one_hot, template)
            loss = self.loss_fn(pred, gt)
        loss["loss"].backward()
        # mse_error (lightning_model.py:119-125) == rec_loss / 3 for 3-vectors: no extra pass over pred/gt
        loss["err"] = loss["rec_loss"].detach() / 3.0
        return loss
