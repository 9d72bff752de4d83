"""The training step as ONE C call: ``PlannedStep`` records the launch sequence of ``training_step`` into a C-side
launch plan (``ops.Plan`` -> pai_plan_*, include/pai_hip.h) and replays it.

The reference's step is one Python call dispatching ~170 ATen operators (reference models/wrapper.py:117-162).  The
eager step of this package issues ~205 kernel launches one by one from Python through ctypes: 5.5 ms of host time under
a 6.4 ms GPU step at batch 64 (round 3), host-bound outright for the smaller per-rank batches of strong scaling.  A
hipGraph of the step (``graph.GraphedStep``, rounds 2-4; removed in round 5) replayed 4-6 % SLOWER than eager issue, because the replay serialises the
weight-gradient stream.  A launch plan keeps the eager schedule exactly -- same kernels, same arguments, same three
streams, same fork / join edges -- and only moves the issue loop from Python into ``pai_plan_run``.

What a plan freezes, and how each piece stays valid:
  * device pointers: engine buffers are persistent (slot pools, arenas, packs); everything the recorded step allocates
    through torch (predictions, logits, loss scalars, gradients w.r.t. logits ...) is kept alive by the plan
    (a ``TorchDispatchMode`` holds every tensor created while recording), so replays reuse those addresses;
  * the bf16 filter packs are double-buffered (the streamed Adam update writes the spare set, ``step()`` swaps): the
    buffer roles alternate from step to step, so there is one plan per buffer-role signature (two in steady state);
  * the Adam step count: recorded Adam launches re-derive their two step-dependent arguments from
    ``step0 + step_delta`` (``pai_plan_run(plan, step_delta)``), the host keeps counting;
  * host-side bookkeeping of a step (optimizer step counts, weight generations, pack commits, logged scalars) is
    replayed in Python after the C call -- a few dozen attribute updates;
  * data parallel: the bucketed all-reduce of ``dist.GradReducer`` through torch.distributed is not a launch of this
    library.  Each collective (and the wait on it) is a HOST node (``host_op``): the recording is split into C-side
    segments around it and the replay alternates ``pai_plan_run`` with those ~2 x buckets Python calls.  With the
    C-ABI communicator (PAI_COMM=rccl, ``pai_allreduce``) the collectives are plan nodes themselves.
The op-level networks of nnops.py (residual / Trans U-Nets) plan the same way: every launch of theirs is a library launch
(gradient fan-in through ``nnops.Fork``, filter layout and patch rearrangement kernels), ``optim.MultiAdam`` is plan-aware.
Anything the recorder cannot own makes it refuse (``disabled`` says why) and the step runs eagerly: kernels launched by
torch itself inside the step (a plug-in ``unet`` module, ``torch.cat`` of the pre-activation residual blocks, loss types
other than "gan"), dropout masks drawn on the host side, per-launch profiling.
"""
from __future__ import annotations

import os

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import ops

# aten operators that launch no kernel (metadata / allocation only); everything else seen while recording is a kernel
# the plan would not contain
_NO_KERNEL = {
    "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape", "alias",
    "detach", "detach_", "as_strided", "permute", "transpose", "t", "slice", "select", "expand", "squeeze", "unsqueeze",
    "unbind", "split", "split_with_sizes", "narrow", "contiguous", "_reshape_alias", "view_as", "lift_fresh",
    "is_pinned", "record_stream", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "size", "stride",
    "numel", "dim", "is_contiguous", "unfold", "diagonal", "flatten", "unflatten", "movedim", "chunk", "set_",
    "result_type", "_has_compatible_shallow_copy_type", "is_same_size", "is_nonzero", "_to_copy_noop",
}


_ACTIVE = None      # the _Recorder of the step being recorded (process-wide: engine hooks run on autograd's thread)


def host_op(fn):
    """Run ``fn()`` now; while a step is being recorded, also make it a HOST node of the plan: the C-side segment
    recorded so far is closed, ``fn`` is replayed from Python between it and the next segment, under the stream that is
    current now.  For the few things a step does that are not launches of libpai_hip.so and cannot be: collectives of
    torch.distributed (dist.GradReducer) and the waits on them.  ``fn`` must only touch buffers that outlive the plan."""
    rec = _ACTIVE
    if rec is None:
        return fn()
    return rec.host_op(fn)


class _Recorder(TorchDispatchMode):
    """Holds every tensor created while a plan is recorded, lists the aten kernels torch launched itself, and splits
    the recording into C-side segments around host nodes (``host_op``)."""

    def __init__(self):
        super().__init__()
        self.keep = []
        self.foreign = []
        self.items = []         # ("plan", ops.Plan) | ("host", fn, stream)
        self.cur = None
        self.exempt = 0

    # ---- segments ----------------------------------------------------------------------------------------
    def begin(self):
        self.cur = ops.Plan()
        self.cur.begin()

    def _close(self):
        if self.cur is not None:
            self.cur.end()
            if self.cur.info()["launches"] + self.cur.info()["waits"] > 0:
                self.items.append(("plan", self.cur))
            self.cur = None

    def end(self):
        self._close()

    def host_op(self, fn):
        stream = torch.cuda.current_stream()
        self._close()
        self.exempt += 1
        try:
            out = fn()
        finally:
            self.exempt -= 1
            self.items.append(("host", fn, stream))
            self.begin()
        return out

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        launches = name not in _NO_KERNEL and not self.exempt
        if launches and name in ("_to_copy", "to", "clone", "contiguous"):
            # a dtype / layout "conversion" that returned its argument launched nothing
            launches = not (torch.is_tensor(out) and args and torch.is_tensor(args[0]) and out.data_ptr() == args[0].data_ptr())
        touched = [t for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t)]
        if launches and any(t.is_cuda for t in touched + [a for a in args if torch.is_tensor(a)]):
            self.foreign.append(name)
        for t in touched:
            if t.is_cuda:
                # the STORAGE is what replays need alive; holding the tensor itself would raise its use count, and autograd's
                # AccumulateGrad then copies a gradient (a kernel of torch's) instead of adopting it as .grad
                self.keep.append(t.untyped_storage())
        return out


class _Recorded:
    __slots__ = ("items", "keep", "logs", "step0", "commits", "opt_steps", "static", "last_use")

    def info(self):
        out = {"launches": 0, "waits": 0, "streams": 0, "runs": 0, "segments": 0, "host_nodes": 0}
        for it in self.items:
            if it[0] == "plan":
                i = it[1].info()
                out["launches"] += i["launches"]
                out["waits"] += i["waits"]
                out["streams"] = max(out["streams"], i["streams"])
                out["runs"] = i["runs"]
                out["segments"] += 1
            else:
                out["host_nodes"] += 1
        return out


class PlannedStep:
    """``step = PlannedStep(model); step(batch, batch_idx)`` in place of ``model.training_step(batch, batch_idx)``.

    The first ``warmup`` calls run eagerly (lazy allocations, parameter adoption by the arena optimizers, the first
    streamed update).  From then on a call whose state signature has a plan replays it; one that has none records one
    (that call executes normally while being recorded)."""

    def __init__(self, model, warmup: int = 3):
        self.model = model
        self.warmup = max(int(warmup), 3)
        self.calls = 0
        self.plans = {}
        self.disabled = None
        self.replays = 0
        self.records = 0
        self.max_plans = 8
        self.evictions = 0

    # ---- eligibility ------------------------------------------------------------------------------------
    def _why_not(self, batch):
        m = self.model
        if not all(torch.is_tensor(b) and b.is_cuda for b in batch):
            return "batch is not on the GPU"
        if not m.training:
            return "model is in eval mode"
        if ops.PROFILE is not None or ops.PROFILE_HBM is not None:
            return "per-launch event profiling is on"
        tr = getattr(m, "trainer", None)
        red = getattr(tr, "reducer", None) if tr is not None else None
        if red is not None and not red.plannable():
            return "the gradient exchange covers parameters outside the engines' arenas (torch-launched staging kernels)"
        for opt in m._all_optimizers():
            if not hasattr(opt, "plan_state"):
                return f"{type(opt).__name__} is not a plan-aware optimizer"
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d) and mod.p > 0:
                return "Dropout2d masks are drawn per step outside the C ABI"
            d = getattr(mod, "dropout", 0)
            if isinstance(d, float) and d > 0:
                return "dropout > 0"
        return None

    # ---- state signature ------------------------------------------------------------------------------------
    def _signature(self, batch):
        m = self.model
        sig = [tuple(b.shape) for b in batch]
        sig.append(torch.cuda.current_stream().cuda_stream)
        for opt in m._all_optimizers():
            sig.append(opt.plan_state())
        h = ops.handle_for(batch[0].device)
        sig.append(tuple(t.data_ptr() if t is not None else 0 for t in (h.workspace, h.scratch, h.wgrad_workspace)))
        return tuple(sig)

    # ---- the call -------------------------------------------------------------------------------------------
    def __call__(self, batch, batch_idx=0):
        m = self.model
        self.calls += 1
        if self.disabled is None and not self.plans:
            self.disabled = self._why_not(batch)
        if self.disabled is not None or self.calls <= self.warmup:
            return m.training_step(batch, batch_idx)
        # conditions that may change between calls (profiling switched on, eval mode, a host batch): this call runs eagerly,
        # the recorded plans stay
        if ops.PROFILE is not None or ops.PROFILE_HBM is not None or not m.training or \
                not all(torch.is_tensor(b) and b.is_cuda for b in batch):
            return m.training_step(batch, batch_idx)
        sig = self._signature(batch)
        if sig is None or any(s is None for s in sig):
            return m.training_step(batch, batch_idx)       # an optimizer is not in its steady state yet
        rec = self.plans.get(sig)
        if rec is None:
            if len(self.plans) >= self.max_plans:
                # a learning-rate schedule (the rate is part of the signature) or a re-registered workspace retires old
                # states for good: drop the least recently used plan -- with the activations and static batches it holds --
                # instead of giving up
                oldest = min(self.plans, key=lambda k: self.plans[k].last_use)
                del self.plans[oldest]
                self.evictions += 1
            return self._record(sig, batch, batch_idx)
        rec.last_use = self.calls
        if not self._replayable(rec):
            # the optimizers' step counts moved apart from the recorded ones (a state reload): the plans are stale
            self.plans.clear()
            return m.training_step(batch, batch_idx)
        return self._replay(rec, batch)

    def _total_steps(self):
        return tuple(opt.total_steps for opt in self.model._all_optimizers())

    def _record(self, sig, batch, batch_idx):
        m = self.model
        rec = _Recorded()
        rec.static = tuple(torch.empty_like(b) for b in batch)
        for s, b in zip(rec.static, batch):
            s.copy_(b)
        rec.step0 = self._total_steps()
        before = m._pai_opt_steps
        logs, orig_log = [], m.log

        def record_log(name, value, *a, **k):
            logs.append((name, value.detach() if torch.is_tensor(value) else value, a, k))
            return orig_log(name, value, *a, **k)
        m.log = record_log
        global _ACTIVE
        recorder = _Recorder()
        try:
            with recorder:
                recorder.begin()
                _ACTIVE = recorder
                try:
                    out = m.training_step(rec.static, batch_idx)
                finally:
                    _ACTIVE = None
                    recorder.end()
        finally:
            del m.log
        rec.items = recorder.items
        rec.keep = recorder.keep
        rec.logs = logs
        rec.opt_steps = m._pai_opt_steps - before
        rec.commits = [opt.plan_commits() for opt in m._all_optimizers()]
        self.records += 1
        if recorder.foreign:
            kinds = sorted(set(recorder.foreign))
            self.disabled = ("the step launches kernels outside the C ABI (" + ", ".join(kinds[:8]) +
                             (", ..." if len(kinds) > 8 else "") + "): it cannot be replayed from a plan")
            self.plans.clear()
            return out
        after = self._total_steps()
        if any(b - a != 1 for a, b in zip(rec.step0, after)):
            self.disabled = "an optimizer did not take exactly one step in the recorded call"
            self.plans.clear()
            return out
        rec.last_use = self.calls
        self.plans[sig] = rec
        return out

    def _replayable(self, rec) -> bool:
        now = self._total_steps()
        delta = now[0] - rec.step0[0]
        return delta >= 0 and all(n - s0 == delta for n, s0 in zip(now, rec.step0))

    def _replay(self, rec, batch):
        m = self.model
        for s, b in zip(rec.static, batch):
            if s.data_ptr() != b.data_ptr():
                s.copy_(b, non_blocking=True)
        now = self._total_steps()
        delta = now[0] - rec.step0[0]
        if any(n - s0 != delta for n, s0 in zip(now, rec.step0)) or delta < 0:
            raise ops.PaiError("PlannedStep: the optimizers' step counts moved apart since the plan was recorded")
        for it in rec.items:
            if it[0] == "plan":
                it[1].run(delta)
            elif it[2].cuda_stream == torch.cuda.current_stream().cuda_stream:
                it[1]()
            else:
                with torch.cuda.stream(it[2]):
                    it[1]()
        # the host-side bookkeeping of the step, as the eager call leaves it
        for opt, commits in zip(m._all_optimizers(), rec.commits):
            opt.plan_replayed(commits)
        m._pai_opt_steps += rec.opt_steps
        for name, value, a, k in rec.logs:
            m.log(name, value, *a, **k)
        self.replays += 1
        return None

    def describe(self) -> dict:
        return {"plans": len(self.plans), "records": self.records, "replays": self.replays, "evictions": self.evictions,
                "disabled": self.disabled,
                "nodes": [r.info() for r in self.plans.values()]}


def enabled_by_default() -> bool:
    """PAI_PLAN=0 turns the planned step off wherever it is the default (Trainer.fit, bench.py)."""
    return os.environ.get("PAI_PLAN", "1") not in ("", "0")
