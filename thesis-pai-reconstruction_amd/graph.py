"""The training step as ONE hipGraph launch.

``UnetWrapper.training_step`` issues ~225 kernel launches from Python (3.9 ms of host time per step at batch 64 on the
reference's GAN step, models/wrapper.py:117-162).  ``GraphedStep`` runs a few steps eagerly (allocations, lazy stream /
attribute set-up, clock ramp), then captures one step -- all three streams of the engine included -- into a
``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) and from then on replays it: one launch per step, the batch copied into
static input tensors first.  Everything step-dependent lives on the device: the Adam step count
(``ArenaAdam.enable_device_step`` -> ``pai_adam_dev``), BatchNorm's ``num_batches_tracked``, the logged scalars
(tensors of the graph's memory pool that every replay overwrites).

Limits: fixed batch shape (a ragged last batch runs eagerly), ``dropout == 0`` (masks are drawn by host-side RNG
plumbing), one process (the bucketed RCCL exchange of ``dist.GradReducer`` is not captured).
"""
from __future__ import annotations

import torch

from . import ops


class GraphedStep:
    def __init__(self, model, warmup: int = 3):
        self.model = model
        self.warmup = max(int(warmup), 1)
        self.calls = 0
        self.graph = None
        self.static = None
        self.shape = None
        self.opt_steps_per_replay = 0
        self.logs = []                # (name, tensor) pairs the captured step passed to self.log()
        # Eager warm-up steps AND the capture run on this side stream: autograd's AccumulateGrad nodes (the composable
        # networks of nnops.py receive their parameter gradients through them) remember the stream they were created on,
        # and nodes created on the default stream make the backward pass of the capture synchronise with it -- an
        # illegal operation during capture (the process aborts).  torch's own recipe: warm up on the capture stream.
        self.stream = None
        self.handle_epoch = -1
        self.disabled = None          # reason string when capture is not possible

    # ---- eligibility ------------------------------------------------------------------------------------
    def _why_not(self, batch):
        m = self.model
        if not all(torch.is_tensor(b) and b.is_cuda for b in batch):
            return "batch is not on the GPU"
        tr = getattr(m, "trainer", None)
        if tr is not None and getattr(tr, "reducer", None) is not None:
            return "data-parallel gradient exchange is not captured"
        if ops.PROFILE is not None:
            return "per-launch event profiling is on"
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d) and mod.p > 0 and m.training:
                return "Dropout2d masks come from host-side RNG plumbing"
            if getattr(mod, "dropout", 0) and isinstance(getattr(mod, "dropout"), float) and mod.dropout > 0 and m.training:
                return "dropout > 0"
        return None

    # ---- the call -------------------------------------------------------------------------------------------
    def __call__(self, batch, batch_idx=0):
        m = self.model
        if self.disabled is None and self.graph is None:
            why = self._why_not(batch)
            if why is not None:
                self.disabled = why
        shape = tuple(tuple(b.shape) for b in batch)
        if self.disabled is not None or (self.shape is not None and shape != self.shape):
            return m.training_step(batch, batch_idx)
        self.calls += 1
        if self.graph is None:
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            if self.calls <= self.warmup:
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    out = m.training_step(batch, batch_idx)
                torch.cuda.current_stream().wait_stream(self.stream)
                return out
            eager = self._capture(batch)
            if self.graph is None:       # capture refused (self.disabled says why): that call ran the step eagerly
                return eager
        if ops.handle_for(batch[0].device).epoch != self.handle_epoch:
            # a workspace / scratch / weight-gradient slab of the handle was re-registered (a validation pass at a larger
            # batch, another layer's plan): the captured kernel arguments point at freed memory
            raise ops.PaiError("GraphedStep: the device handle's buffers were replaced after the capture; the graph is stale "
                               "(size the buffers before capturing, or build a new GraphedStep)")
        for s, b in zip(self.static, batch):
            s.copy_(b, non_blocking=True)
        self.graph.replay()
        for name, value in self.logs:        # the logged scalars are tensors of the graph's pool: same objects, new values
            m.log(name, value)
        m._pai_opt_steps += self.opt_steps_per_replay
        for opt in m._all_optimizers():
            if hasattr(opt, "note_replays"):
                opt.note_replays(1)
        return None

    def _capture(self, batch):
        m = self.model
        opts = m._all_optimizers()
        for opt in opts:       # all or nothing: no optimizer is switched to device-side counting unless every one can be
            if not hasattr(opt, "enable_device_step"):
                self.disabled = f"{type(opt).__name__} keeps its step count on the host"
                return m.training_step(batch, 0)
        for opt in opts:
            opt.enable_device_step()
        # the engines' cross-stream events recorded by earlier (un-captured) steps must not be waited on inside the capture
        for mod in m.modules():
            eng = getattr(mod, "engine", None) if hasattr(type(mod), "engine") else None
            side = getattr(eng, "_side", None)
            if side is not None:
                side._scratch_marked = False
        self.static = tuple(torch.empty_like(b) for b in batch)
        for s, b in zip(self.static, batch):
            s.copy_(b)
        self.shape = tuple(tuple(b.shape) for b in batch)
        torch.cuda.synchronize()
        before = m._pai_opt_steps
        steps_before = [o._arena_steps for o in opts]
        g = torch.cuda.CUDAGraph()
        logs, orig_log = [], m.log

        def record(name, value, *a, **k):
            logs.append((name, value.detach() if torch.is_tensor(value) else value))
        m.log = record
        try:
            with torch.cuda.graph(g, stream=self.stream):
                m.training_step(self.static, 0)
        finally:
            del m.log                 # back to the class's method
            assert m.log.__func__ is orig_log.__func__
        self.logs = logs
        self.handle_epoch = ops.handle_for(batch[0].device).epoch
        # the capture itself executed nothing: undo the host-side bookkeeping of the captured step() calls
        self.opt_steps_per_replay = m._pai_opt_steps - before
        m._pai_opt_steps = before
        for o, n in zip(opts, steps_before):
            o._arena_steps = n
        self.graph = g
