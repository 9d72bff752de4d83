"""``ArenaAdam``: torch.optim.Adam whose step is one fused kernel over flat arenas.

The reference builds two stock ``torch.optim.Adam`` objects (models/wrapper.py:97-115).  For the
networks that run on the HIP engines the parameters, their gradients and both Adam moments live in
flat fp32 buffers with one common layout (engine.GradArena), so the whole update is a single
HBM-bound pass (`pai_adam`, 28 B per parameter) instead of ~10 multi-tensor passes.  The object is
still a ``torch.optim.Adam``: ``param_groups`` (used by toggle_optimizer), ``state_dict`` and the
update rule (no weight decay / amsgrad) are unchanged, and whenever the arena preconditions do not
hold (foreign gradients, parameters moved, CPU) it falls back to the stock implementation with the
same moment tensors.
"""
from __future__ import annotations

import os

import torch

from . import ops


class ArenaAdam(torch.optim.Adam):
    def __init__(self, params, engine, lr=2e-4, betas=(0.5, 0.999), eps=1e-7):
        super().__init__(list(params), lr=lr, betas=betas, eps=eps, foreach=True)
        self._engine = engine
        self._arena_steps = 0          # steps taken on the fused path and not yet mirrored into `state`
        self._dev_step = None          # device-step mode: int64 device scalar holding the step count (a caller's hipGraph capture of step())
        self._dev_coeff = None
        self._streamed = None          # streaming step armed: elements of the arena already updated by the hook
        self._stream_step = 0
        self._reducer = None           # the GradReducer the armed step subscribed to (data parallel)
        # (Streamed ranges always run on the stream that produced their gradients.  A stream of their own was measured in
        #  round 3 -- 7.17-7.27 ms/step against 6.31-6.35: a third concurrent stream takes CUs and HBM from two matrix-bound
        #  ones -- and removed in round 5: its two torch-level stream edges were not nodes of a launch plan.)
        # PAI_ADAM_PACK=0: never write filter packs from the streamed update (A/B switch; default on)
        self._fuse_packs = os.environ.get("PAI_ADAM_PACK", "1") not in ("", "0")
        self._pack_targets = None      # armed step without a reducer: the engine's pack_targets()
        self._packs_written = []       # _Packs whose spare buffers the armed step has filled; committed by step()
        self._last_commits = []        # ... of the most recent step() (plan_commits)

    # ---- hipGraph support -----------------------------------------------------------------------------
    def enable_device_step(self):
        """Keep the step count on the device from now on (pai_adam_dev): needed when ``step()`` is captured into a
        hipGraph, where a host-side count would be frozen into the captured kernel arguments.  ``note_replays(n)``
        tells the host-side bookkeeping (state_dict, total_steps) about n replays of a graph containing one step."""
        if self._dev_step is None:
            dev = self.param_groups[0]["params"][0].device
            self._dev_step = torch.tensor(self.total_steps, dtype=torch.int64, device=dev)
            self._dev_coeff = torch.zeros(2, dtype=torch.float32, device=dev)

    def note_replays(self, n: int = 1):
        self._arena_steps += n

    # ---- launch-plan support (plan.PlannedStep) -----------------------------------------------------------------
    def plan_state(self):
        """Hashable signature of everything a recorded training step froze about this optimizer and its engine: the
        update rule's constants, the arenas and WHICH of the double-buffered filter packs is current (the streamed update
        writes the spare set and ``step()`` swaps, so the roles alternate from step to step).  None while the optimizer
        is not in its steady state (parameters not adopted by the arena yet, packs stale, a device-side step count)."""
        eng = self._engine
        arena = eng.arena()
        if self._dev_step is not None or self._streamed is not None or not arena.params_adopted():
            return None
        if len(self.param_groups) != 1:
            return None
        g = self.param_groups[0]
        ptrs = []
        for pk in eng.all_packs():
            if pk.wf is None or pk.dtype is None:
                return None
            ptrs.append(pk.wf.data_ptr())
        mode, _ = self._stream_mode()
        if mode == "hook" and self._fuse_packs:
            # the streamed update writes the packs itself: in the steady state none is stale when a step starts (a stale
            # one means the previous step went another way -- the next forward re-packs, and THAT step must not be the
            # recorded one).  Without the streamed update every step starts stale and re-packs: that IS its steady state.
            for t in eng.pack_targets():
                if t[5].dtype == torch.bfloat16 and t[5]._stale(t[5].dtype):      # fp32: the parameter IS the pack
                    return None
        return (mode, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), arena.flat.data_ptr(),
                arena.pflat.data_ptr(), tuple(ptrs))

    def plan_commits(self):
        """The layers whose spare packs the LAST ``step()`` committed (what a replay of that step has to commit too)."""
        return list(self._last_commits)

    def plan_replayed(self, commits):
        """Host-side bookkeeping of one replayed step: what ``step()`` does besides launching."""
        self._arena_steps += 1
        self._engine.weights_generation[0] += 1
        for pk in commits:
            pk.commit_replayed()

    # ---- streaming step -------------------------------------------------------------------------------
    def arm_streaming(self) -> bool:
        """Let the NEXT backward pass of this optimizer's network start the update while it is still running: the
        engine reports every range of the gradient arena the moment it is final (the hook data-parallel buckets hang
        on), and that range's Adam launch goes out right there, on the stream that produced the last gradient of the
        range.  Adam is a 28 B/parameter HBM stream and the backward pass is matrix-bound, so the two overlap well
        (the 54 M-parameter generator: 250 us of a 7 ms step).  Same arithmetic, same step count: the result is the
        one ``step()`` alone would give; ``step()`` then only updates what the hook has not reached, bumps the
        bookkeeping and disarms.  Returns False (and changes nothing) when something else owns the hook (a gradient
        reducer: the all-reduce has to come first), when the step count lives on the device (graph capture) or the
        arena preconditions do not hold."""
        mode, reducer = self._stream_mode()
        if mode is None:
            return False
        arena = self._engine.arena()
        self._pack_targets = None
        if reducer is not None:
            if not reducer.subscribe(arena, self._on_reduced):
                return False
            self._reducer = reducer
        else:
            self._engine.grad_ready_hook = self._on_ready
            targets = getattr(self._engine, "pack_targets", None)
            self._pack_targets = targets() if (targets is not None and self._fuse_packs) else None
        self._packs_written = []
        self._streamed = 0
        self._stream_step = self.total_steps + 1
        return True

    def _stream_mode(self):
        """(mode, reducer): whether ``arm_streaming`` would arm now -- "hook" (this optimizer takes the engine's
        gradient-ready hook), "reducer" (it subscribes to the GradReducer that owns the hook) or None."""
        if os.environ.get("PAI_NO_STREAM_ADAM", "0") not in ("", "0"):      # A/B switch
            return None, None
        hook = getattr(self._engine, "grad_ready_hook", None)
        if hook is not None and getattr(hook, "__func__", None) is getattr(self._on_ready, "__func__", object()) \
                and getattr(hook, "__self__", None) is self:
            hook = None            # our own hook, left armed: same as free
        reducer = getattr(hook, "__self__", None) if hook is not None else None
        if self._dev_step is not None or (hook is not None and not hasattr(reducer, "subscribe")):
            return None, None
        group = self.param_groups[0]
        if len(self.param_groups) != 1 or group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
            return None, None
        params = group["params"]
        if not params or not params[0].is_cuda:
            return None, None
        arena = self._engine.arena()
        # only once the parameters already LIVE in the arena (the first fused step() moves them there): moving them
        # between a forward pass and its backward pass would invalidate the filter packs that forward made
        if len(params) != len(arena.params) or not arena.params_adopted():
            return None, None
        if reducer is not None:
            # data parallel: the reducer owns the hook; the update of a bucket follows its all-reduce + average.
            # OPT-IN (PAI_DDP_STREAM_ADAM=1): this path updates parameters on a third stream while the backward pass
            # is still running and has only been exercised with two ranks on ONE GPU over gloo (tests/test_gpu_ddp.py),
            # never over RCCL on several GPUs; until it has, the default under a reducer is the update in step().
            if os.environ.get("PAI_DDP_STREAM_ADAM", "0") in ("", "0") or reducer.world < 2:
                return None, None
            return "reducer", reducer
        return "hook", None

    def _adam_range(self, arena, a, b, step):
        group = self.param_groups[0]
        if b > a:
            ops.adam(arena.pflat[a:b], arena.flat[a:b], arena.mflat[a:b], arena.vflat[a:b], float(group["lr"]),
                     float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), step)

    def _adam_range_packing(self, arena, a, b, step):
        """``_adam_range`` whose launches also write the bf16 filter packs of the dense conv weights inside [a, b)
        (``pai_adam_pack``), into the layers' SPARE pack buffers: the input-gradient kernels of the running backward pass
        still read the current ones.  ``step()`` commits them."""
        inside = [t for t in (self._pack_targets or ()) if a <= t[0] and t[0] + t[1] <= b]
        group = self.param_groups[0]
        cur = a
        for k, (o, n, cout, taps, cin, pk) in enumerate(inside):
            spare = pk.spare()
            if spare is None:
                continue
            end = inside[k + 1][0] if k + 1 < len(inside) else b
            ops.adam_pack(arena.pflat[cur:end], arena.flat[cur:end], arena.mflat[cur:end], arena.vflat[cur:end], o - cur,
                          cout, taps, cin, spare[0], spare[1], float(group["lr"]), float(group["betas"][0]),
                          float(group["betas"][1]), float(group["eps"]), step)
            self._packs_written.append(pk)
            cur = end
        self._adam_range(arena, cur, b, step)

    def _on_ready(self, arena, end_offset):
        if int(end_offset) < self._streamed:
            # a SECOND backward pass into this arena before step(): its gradients would be added behind an update that
            # has already consumed the first pass (GradArena.begin_backward allows accumulation; a streaming step
            # cannot honour it) -- same guard as GradReducer._on_ready
            raise ops.PaiError("ArenaAdam: a streaming step is armed and a second backward pass reached this arena "
                               "before step(); use one backward pass per optimizer step or do not arm_streaming()")
        with torch.no_grad():
            # (round 6 re-measured the update of a range on a stream of its own, ordered behind the launch that made the
            #  range final: 6.66 against 5.93 ms/step -- a third concurrent stream of light workgroups takes the dispatch slots
            #  of the two matrix-bound ones, as in round 3; the update stays on the stream that produced the gradient)
            self._adam_range_packing(arena, self._streamed, int(end_offset), self._stream_step)
        self._streamed = max(self._streamed, int(end_offset))

    def _on_reduced(self, arena, lo, hi):
        if lo != self._streamed:
            raise ops.PaiError(f"ArenaAdam: reduced bucket [{lo}, {hi}) does not continue the updated range "
                               f"[0, {self._streamed})")
        with torch.no_grad():
            self._adam_range(arena, lo, hi, self._stream_step)
        self._streamed = hi

    def _disarm(self):
        if getattr(self._engine, "grad_ready_hook", None) == self._on_ready:
            self._engine.grad_ready_hook = None
        reducer = getattr(self, "_reducer", None)
        if reducer is not None:
            # a subscription whose backward pass never activated the arena must not fire in a later pass
            reducer.unsubscribe(self._engine.arena())
            self._reducer = None
        self._streamed = None

    # ---- helpers --------------------------------------------------------------------------------
    def _arena_ready(self):
        group = self.param_groups[0]
        if len(self.param_groups) != 1 or group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
            return None
        params = group["params"]
        if not params or not params[0].is_cuda:
            return None
        arena = self._engine.arena()
        if len(params) != len(arena.params) or not arena.adopt_parameters():
            return None
        for p in params:
            if p.grad is None or p.grad.data_ptr() != arena.view(p).data_ptr():
                return None
        return arena

    def _mirror_state(self, arena):
        """Expose the arena moments through the regular per-parameter optimizer state."""
        for p in self.param_groups[0]["params"]:
            st = self.state[p]
            if "exp_avg" not in st:
                m, v = arena.moment_views(p)
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"], st["exp_avg_sq"] = m, v
            st["step"] += self._arena_steps
        self._arena_steps = 0

    @property
    def total_steps(self) -> int:
        any_state = next(iter(self.state.values()), None)
        done = int(any_state["step"]) if any_state and "step" in any_state else 0
        return done + self._arena_steps

    # ---- optimizer API ------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        streamed = self._streamed
        if streamed is not None:
            self._disarm()
        self._last_commits = []
        arena = self._arena_ready()
        if arena is None:
            if streamed:
                raise ops.PaiError("ArenaAdam: a streaming step was armed and partly applied, but the gradients are "
                                   "not the arena's any more")
            if self._arena_steps:
                self._mirror_state(self._engine.arena())
            return super().step(closure)
        group = self.param_groups[0]
        if streamed is not None:
            self._adam_range(arena, streamed, arena.flat.numel(), self._stream_step)     # what the hook did not reach
            self._arena_steps += 1
            self._engine.weights_generation[0] += 1
            for pk in self._packs_written:      # their spare buffers hold the packs of the weights as they are now
                pk.commit()
            self._last_commits = self._packs_written
            self._packs_written = []
            return None
        if self._dev_step is not None:
            ops.adam_dev(arena.pflat, arena.flat, arena.mflat, arena.vflat, float(group["lr"]), float(group["betas"][0]),
                         float(group["betas"][1]), float(group["eps"]), self._dev_step, self._dev_coeff)
        else:
            step = self.total_steps + 1
            ops.adam(arena.pflat, arena.flat, arena.mflat, arena.vflat, float(group["lr"]), float(group["betas"][0]),
                     float(group["betas"][1]), float(group["eps"]), step)
        self._arena_steps += 1
        self._engine.weights_generation[0] += 1   # master weights changed behind torch's version counters
        return None

    def state_dict(self):
        if self._arena_steps:
            self._mirror_state(self._engine.arena())
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Restore moments and step count.  On the fused path the moments LIVE in the arena (``mflat`` / ``vflat``):
        the loaded tensors are copied into the arena views and the per-parameter state is re-pointed at those views, so
        that the next fused step continues from the loaded moments and bias correction instead of from zero."""
        if self._arena_steps:
            self._mirror_state(self._engine.arena())
        super().load_state_dict(state_dict)
        params = self.param_groups[0]["params"]
        if not params or not params[0].is_cuda:
            return
        arena = self._engine.arena()
        if len(params) != len(arena.params) or not arena.adopt_parameters():
            return
        with torch.no_grad():
            for p in params:
                st = self.state.get(p)
                if not st or "exp_avg" not in st:
                    continue
                m, v = arena.moment_views(p)
                m.copy_(st["exp_avg"].to(m.device).reshape(m.shape))
                v.copy_(st["exp_avg_sq"].to(v.device).reshape(v.shape))
                st["exp_avg"], st["exp_avg_sq"] = m, v
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32)   # host scalar, as _mirror_state keeps it
        self._arena_steps = 0
        if self._dev_step is not None:
            # graph mode: the bias correction continues from the LOADED step count, not from the one captured before
            self._dev_step.fill_(self.total_steps)


class MultiAdam(torch.optim.Adam):
    """torch.optim.Adam whose step is ``pai_adam_multi``: the same fused update as ``ArenaAdam`` for networks whose
    parameters are separate allocations (the composable residual / Trans U-Nets) -- one 28 B/parameter pass in a few
    launches instead of the ~10 multi-tensor passes of the stock foreach implementation (17 ms -> 5 ms per step on the
    1.03 B-parameter TransUNet).  State layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter) is the stock one;
    anything outside the preconditions (CPU tensors, sparse / non-fp32 gradients, options other than the reference's)
    goes to the stock implementation."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-7):
        super().__init__(list(params), lr=lr, betas=betas, eps=eps, foreach=True)
        self._dev_step = None          # device-step mode: int64 device scalar holding the step count
        self._dev_coeff = None
        self._arena_steps = 0          # replays not yet mirrored into the per-parameter `step` entries

    # ---- hipGraph support (same contract as ArenaAdam) ----------------------------------------------------------
    def _total_steps(self) -> int:
        st = next(iter(self.state.values()), None)
        return (int(st["step"]) if st and "step" in st else 0) + self._arena_steps

    def enable_device_step(self):
        if self._dev_step is None:
            dev = self.param_groups[0]["params"][0].device
            self._dev_step = torch.tensor(self._total_steps(), dtype=torch.int64, device=dev)
            self._dev_coeff = torch.zeros(2, dtype=torch.float32, device=dev)

    def note_replays(self, n: int = 1):
        self._arena_steps += n

    # ---- launch plans (plan.PlannedStep; same contract as ArenaAdam) ---------------------------------------------
    @property
    def total_steps(self) -> int:
        return self._total_steps()

    def plan_state(self):
        """What a recorded step bakes in besides pointers that never move (parameters, moments): the hyper-parameters passed
        by value.  None until the moments exist (first step) or while the fused update's preconditions do not hold."""
        if len(self.param_groups) != 1 or self._dev_step is not None:
            return None
        group = self.param_groups[0]
        if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
            return None
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps or any("exp_avg" not in self.state.get(p, ()) for p in ps):
            return None
        return ("multi", float(group["lr"]), tuple(float(b) for b in group["betas"]), float(group["eps"]), len(ps))

    def plan_commits(self):
        """The gradient tensors the recorded step produced: replays write the same memory."""
        return [(p, p.grad) for p in self.param_groups[0]["params"] if p.grad is not None]

    def plan_replayed(self, commits):
        self._arena_steps += 1
        for p, g in commits:
            if p.grad is not g:
                p.grad = g

    def _mirror_steps(self):
        if self._arena_steps:
            for st in self.state.values():
                if "step" in st:
                    st["step"] += self._arena_steps
            self._arena_steps = 0

    def state_dict(self):
        self._mirror_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._arena_steps = 0
        super().load_state_dict(state_dict)
        if self._dev_step is not None:
            self._dev_step.fill_(self._total_steps())

    @torch.no_grad()
    def step(self, closure=None):
        from . import nnops
        nnops.join_wgrads()      # weight gradients of the composable networks run on a second stream; no-op after manual_backward

        def stock(*a):
            # a captured step (device-side count) must never reach the stock implementation: it would
            # bake the HOST step count of the capture into the graph and every replay would reuse one bias correction
            if self._dev_step is not None:
                raise ops.PaiError("MultiAdam: the fused update's preconditions do not hold (closure / several param groups / "
                                   "weight decay / non-contiguous or non-fp32 gradients / mixed step counts) while the step "
                                   "count lives on the device (captured step): run this model eagerly")
            return torch.optim.Adam.step(self, *a)
        if closure is not None or len(self.param_groups) != 1:
            return stock(closure)
        group = self.param_groups[0]
        if group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"]:
            return stock()
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        if any((not p.is_cuda) or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse
               or not p.is_contiguous() or not p.grad.is_contiguous() for p in ps):
            return stock()
        steps = set()
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["step"].is_cuda:
                return stock()
            steps.add(int(st["step"]))
        if len(steps) != 1:          # parameters with different histories (e.g. unused in some steps): stock path
            return stock()
        if self._dev_step is not None:
            # captured step: the count lives on the device and advances with every replay (note_replays keeps the host
            # bookkeeping in step with it)
            ops.adam_multi_dev([p.data for p in ps], [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                               [self.state[p]["exp_avg_sq"] for p in ps], float(group["lr"]), float(group["betas"][0]),
                               float(group["betas"][1]), float(group["eps"]), self._dev_step, self._dev_coeff)
            self._arena_steps += 1
            return None
        self._mirror_steps()
        step = steps.pop() + 1
        ops.adam_multi([p.data for p in ps], [p.grad for p in ps], [self.state[p]["exp_avg"] for p in ps],
                       [self.state[p]["exp_avg_sq"] for p in ps], float(group["lr"]), float(group["betas"][0]),
                       float(group["betas"][1]), float(group["eps"]), step)
        for p in ps:
            self.state[p]["step"] += 1
        return None


def make_adam(module: torch.nn.Module, lr, betas, eps):
    eng = getattr(module, "engine", None) if hasattr(type(module), "engine") else None
    if eng is not None:
        return ArenaAdam(module.parameters(), eng, lr=lr, betas=betas, eps=eps)
    return MultiAdam(module.parameters(), lr=lr, betas=betas, eps=eps)
