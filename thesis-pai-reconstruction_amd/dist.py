"""Data-parallel gradient averaging: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code (it relies on Lightning's implicit DDP); this is the
MI355X-native equivalent for its training step.  Parameter gradients of the HIP networks live
in flat fp32 arenas ordered by backward-completion time (engine.GradArena).  While the
backward pass is still running, every completed ``bucket_bytes`` range of an arena is handed to
``torch.distributed.all_reduce(async_op=True)`` -- backend "nccl" is RCCL on ROCm -- which
runs on the process group's own stream, so the collective of bucket k overlaps the kernels
that produce bucket k+1.  ``finish()`` (called from ``manual_backward``) sends the tail, waits
for everything and applies the 1/world_size average.  BatchNorm statistics stay local
(non-synchronised), exactly like Lightning-DDP without ``sync_batchnorm``.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _CommWork:
    """Completion handle of a pai_allreduce issued on the communication stream (mirrors torch's Work.wait()); the event
    is an ``ops.Event`` of the reducer's pool, so record and wait are C-ABI calls (nodes of a launch plan)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        self.event.wait(torch.cuda.current_stream())


class _TorchWork:
    """A torch.distributed collective as a pair of HOST nodes of the launch plan (plan.host_op): ``issue`` launches it
    (torch's process-group stream waits for the stream that is current at that point), ``wait`` orders the current
    stream behind it.  Replayed from Python with the same buffers, in the same order, between the C-side segments."""

    def __init__(self, buf, pg):
        from . import plan as _plan
        self._plan = _plan
        self.work = None

        def issue():
            self.work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=pg, async_op=True)
        self._plan.host_op(issue)

    def wait(self):
        self._plan.host_op(lambda: self.work.wait())


def _cast(src, dst):
    """dst <- src across fp32 / bf16: a C-ABI launch on the HIP device (a node of the launch plan), torch's copy_ on CPU."""
    if src.is_cuda:
        from . import ops
        ops.cast(src, dst)
    else:
        dst.copy_(src)


def _cast_many(pairs):
    """_cast for a list of (src, dst) pairs: eight per launch on the HIP device (pai_cast_multi)."""
    if pairs and pairs[0][0].is_cuda:
        from . import ops
        for i in range(0, len(pairs), 8):
            ops.cast_multi(pairs[i:i + 8])
    else:
        for src, dst in pairs:
            dst.copy_(src)


def _scale(t, f):
    if t.is_cuda:
        from . import ops
        ops.scale_(t, f)
    else:
        t.mul_(f)


def _stream_wait(waiting, signalling):
    from . import ops
    ops.stream_wait(waiting, signalling)


RCCL_KNOBS = {"algo": "NCCL_ALGO", "proto": "NCCL_PROTO", "min_channels": "NCCL_MIN_NCHANNELS", "max_channels": "NCCL_MAX_NCHANNELS"}


def configure_rccl(algo=None, proto=None, min_channels=None, max_channels=None):
    """Collective-algorithm knobs of RCCL, to be called BEFORE the communicator exists (``init_from_env`` /
    ``make_rccl_comm``): RCCL reads them when it builds its channels.  ``None`` leaves a knob to RCCL -- the default, and
    what the 228 MB gradient exchange of the Pix2Pix generator should take on a fully connected 8-GPU xGMI node: RCCL's
    topology search lays its ring channels over DIFFERENT Hamiltonian cycles of the 7-link clique, so "Ring" there is a
    multi-link algorithm (all seven links of a GPU carry a share of every bucket; SURVEY section 5's 0.18-0.36 ms figure
    for a direct reduce-scatter + all-gather is its bandwidth term), not the one-link ring of 1.25-2.5 ms.  ``algo``
    ("Ring" | "Tree"), ``proto`` ("Simple" | "LL" | "LL128") and the channel bounds exist for the first hardware run:
    forcing "Tree" or a low channel count is how one would SEE a link-bound exchange in the bench line.  Returns the
    settings in force (also the ones inherited from the environment)."""
    import os
    for key, val in (("algo", algo), ("proto", proto), ("min_channels", min_channels), ("max_channels", max_channels)):
        if val is not None:
            if dist.is_initialized():
                raise RuntimeError("configure_rccl: the process group already exists; RCCL reads its knobs at communicator creation")
            os.environ[RCCL_KNOBS[key]] = str(val)
    return {k: os.environ.get(v) for k, v in RCCL_KNOBS.items()}


def make_rccl_comm(process_group=None):
    """A C-ABI RCCL communicator (ops.Comm: pai_comm_init / pai_allreduce) spanning the ranks of the initialised
    torch.distributed job; the 128-byte id travels through torch.distributed's object broadcast."""
    from . import ops
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    box = [ops.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=process_group)
    return ops.Comm(box[0], rank, world)


class GradReducer:
    """``comm``: an ``ops.Comm`` (RCCL through the C ABI, ``make_rccl_comm()``; also selected by PAI_COMM=rccl) --
    the buckets are then reduced by ``pai_allreduce`` on a communication stream of this object instead of
    ``torch.distributed.all_reduce``; bucketing, overlap and averaging are the same."""

    def __init__(self, process_group=None, bucket_bytes: Optional[int] = None, overlap: bool = True, comm=None,
                 grad_dtype: Optional[torch.dtype] = None):
        """``grad_dtype``: wire format of the arena buckets, torch.float32 (default) or torch.bfloat16 (also
        PAI_GRAD_DTYPE=bf16).  With bf16 every bucket is cast into a persistent bf16 staging arena, reduced there and
        cast back when it is waited for: half the bytes per xGMI link (109 MB instead of 218 MB for the Pix2Pix
        generator) at bf16 rounding of the summands; the master gradients stay fp32."""
        import os
        self.pg = process_group
        # Bucket size (also PAI_DDP_BUCKET_MB).  32 MB: an all-reduce over R = 8 ranks is 2 (R - 1) = 14 chunk hops of
        # B / R bytes; at ~10 us per hop and ~300 GB/s of all-reduce bus bandwidth over seven xGMI links a 32 MB bucket is
        # ~0.14 ms of hop latency + ~0.19 ms of bytes, a 64 MB one 0.14 + 0.37 -- but the 218 MB arena then leaves in four
        # pieces instead of seven and the last one (encoders[0..2], final only at the very end of the backward pass) is the
        # exposed one: smaller tail, more overlap.  Below ~8 MB the hop latency is the whole cost.
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("PAI_DDP_BUCKET_MB", "32")) * (1 << 20))
        if grad_dtype is None:
            grad_dtype = torch.bfloat16 if os.environ.get("PAI_GRAD_DTYPE", "").lower() in ("bf16", "bfloat16") else torch.float32
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"GradReducer: grad_dtype must be float32 or bfloat16, not {grad_dtype}")
        self.grad_dtype = grad_dtype
        self.comm = comm
        if comm is None and os.environ.get("PAI_COMM", "") == "rccl" and dist.is_initialized() and \
                dist.get_world_size(process_group) > 1 and torch.cuda.is_available():
            self.comm = make_rccl_comm(process_group)
        self._comm_stream = None
        self._events, self._ev_next = [], 0
        self.bucket_elems = max(int(bucket_bytes) // 4, 1)
        self.overlap = overlap
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._arenas = {}       # id(arena) -> state
        self._subs = {}         # id(arena) -> callback(arena, lo, hi): run on every bucket once it is reduced + averaged
        self._post_stream = None
        self._engines = []
        self._foreign_params: List[torch.nn.Parameter] = []
        self._foreign_hooks, self._fbuckets, self._fmap = [], [], {}
        self._f32_scratch = None
        self.stats = {"buckets": 0, "bytes": 0}

    def plannable(self) -> bool:
        """plan.PlannedStep: arena buckets and the fixed buckets of the gradients outside the arenas (the composable
        residual / Trans U-Nets) are exchanged with C-ABI launches (cast, scale, pai_allreduce) plus, on the
        torch.distributed path, host nodes for the collectives.  What cannot be replayed is a gradient the staging
        launches cannot read as it stands: not fp32, or not contiguous (``torch`` would copy it first)."""
        return all(p.dtype == torch.float32 and p.is_contiguous() for p in self._foreign_params)

    def describe(self) -> dict:
        """What the exchange of this reducer looks like, for the bench line: transport, ranks, RCCL knobs in force, bucket
        size, wire type, and the buckets / bytes the last step(s) sent (``stats``)."""
        import os
        backend = None
        if dist.is_initialized():
            backend = dist.get_backend(self.pg)
        return {"transport": ("pai_allreduce (C-ABI RCCL communicator)" if self.comm is not None else
                              f"torch.distributed.all_reduce, backend {backend}"),
                "rccl_ranks": self.rccl_ranks(), "world": int(self.world),
                "rccl_knobs": {k: os.environ.get(v) for k, v in RCCL_KNOBS.items()},
                "algorithm": (os.environ.get("NCCL_ALGO") or "RCCL default (topology search; multi-channel rings over all xGMI links)")
                             if self.rccl_ranks() else f"not RCCL ({backend}: host all-reduce; a plumbing run, not a scaling figure)",
                "bucket_bytes": int(self.bucket_elems) * 4, "wire_dtype": str(self.grad_dtype).replace("torch.", ""),
                "overlap_with_backward": bool(self.overlap),
                "buckets_sent": int(self.stats.get("buckets", 0)), "bytes_sent": int(self.stats.get("bytes", 0))}

    def rccl_ranks(self) -> int:
        """Ranks of the RCCL communicator the buckets actually travel over: the C-ABI communicator's own count, or the
        size of the torch.distributed group when its backend is "nccl" (= RCCL on ROCm); 0 when the exchange is not RCCL
        (gloo, one rank)."""
        if self.comm is not None:
            return int(self.comm.world)
        if dist.is_initialized() and dist.get_backend(self.pg) == "nccl":
            return int(dist.get_world_size(self.pg))
        return 0

    # ---- wiring ---------------------------------------------------------------------------
    def attach(self, model: torch.nn.Module):
        """Hook every HIP engine found under ``model``; remember the parameters that are not
        covered by an arena (plug-in ``unet`` modules) for a coalesced reduction in finish()."""
        covered = set()
        for m in model.modules():
            eng = getattr(m, "engine", None) if hasattr(type(m), "engine") else None
            if eng is not None and hasattr(eng, "arena"):
                self.attach_engine(eng)
                covered.update(id(p) for p, _ in eng.ordered_params())
        self._attach_foreign([p for p in model.parameters() if id(p) not in covered])

    # ---- gradients outside the arenas: fixed buckets fed by post-accumulate-grad hooks ---------------
    def _attach_foreign(self, params):
        """Parameters whose gradients are ordinary tensors (the composable residual / Trans U-Nets of ``nnops``, plug-in
        ``unet`` modules).  What Lightning-DDP does for the reference (``main.py:123-136``): fixed buckets in reverse
        registration order -- the order the backward pass finishes them in -- each all-reduced the moment its last
        gradient is final, i.e. beside the rest of the backward pass.  A tensor of at least one bucket is reduced in
        place (bf16 wire: through a persistent staging buffer); smaller ones are gathered into a persistent bucket buffer
        by the hook (``pai_cast``: a node of the launch plan, not ``torch.cat``) and scattered back in ``finish()``."""
        for h in self._foreign_hooks:
            h.remove()
        self._foreign_hooks = []
        self._foreign_params = list(params)
        self._fbuckets, self._fmap = [], {}
        cur = None
        for p in reversed(self._foreign_params):
            if not p.requires_grad:
                continue
            n = p.numel()
            if n >= self.bucket_elems:
                b = {"solo": True, "members": [(p, 0, n)], "numel": n, "buf": None, "work": None, "have": set(), "touched": set()}
                self._fbuckets.append(b)
            else:
                if cur is None or cur["numel"] >= self.bucket_elems:
                    cur = {"solo": False, "members": [], "numel": 0, "buf": None, "work": None, "have": set(), "touched": set()}
                    self._fbuckets.append(cur)
                b = cur
                off = (b["numel"] + 7) // 8 * 8          # 32-byte pieces: the multi-tensor scatter's contract
                b["members"].append((p, off, n))
                b["numel"] = off + n
            self._fmap[id(p)] = (b, len(b["members"]) - 1)
            # a hook keeps the parameter's weight gradient on the main stream (nnops._WgradStream._hooked): only where it
            # buys something -- more than one rank AND buckets that leave during the backward pass.  Otherwise finish()
            # gathers from p.grad (its `not hooks` path).
            if self.world > 1 and self.overlap and hasattr(p, "register_post_accumulate_grad_hook"):
                self._foreign_hooks.append(p.register_post_accumulate_grad_hook(self._foreign_ready))

    def _fbuf(self, b, like, dtype):
        if b["buf"] is None or b["buf"].device != like.device or b["buf"].dtype != dtype:
            b["buf"] = torch.zeros(b["numel"], dtype=dtype, device=like.device)
        return b["buf"]

    def _foreign_ready(self, p):
        """post-accumulate-grad hook: ``p.grad`` is final for this backward pass."""
        ent = self._fmap.get(id(p))
        if self.world < 2 or ent is None or p.grad is None:
            return
        b, k = ent
        b["touched"].add(k)
        if self.overlap:
            self._foreign_stage(b, k)

    def _foreign_stage(self, b, k):
        """Member k of bucket b has its gradient: stage it; all-reduce the bucket when it is complete."""
        if k in b["have"]:
            raise RuntimeError("GradReducer: a second backward pass produced a gradient whose bucket was already "
                               "filled; use one backward per optimizer step")
        b["have"].add(k)
        p, off, n = b["members"][k]
        g = p.grad
        if b["solo"]:
            if self.grad_dtype == torch.float32 and g.dtype == torch.float32 and g.is_contiguous():
                b["work"], b["inplace"] = self._all_reduce_async(g), g
            else:
                stage = self._fbuf(b, g, self.grad_dtype)
                _cast(g.contiguous().view(-1), stage)
                b["work"], b["inplace"] = self._all_reduce_async(stage), None
            self._count(n)
            return
        buf = self._fbuf(b, g, self.grad_dtype)
        _cast(g.contiguous().view(-1), buf[off:off + n])
        if len(b["have"]) == len(b["members"]):
            b["work"] = self._all_reduce_async(buf)
            self._count(b["numel"])

    def _count(self, numel):
        self.stats["buckets"] += 1
        self.stats["foreign_buckets"] = self.stats.get("foreign_buckets", 0) + 1
        self.stats["bytes"] += numel * (4 if self.grad_dtype == torch.float32 else 2)

    def _finish_foreign(self):
        scale = 1.0 / self.world
        # buckets the backward pass left incomplete (some members got no gradient in this pass), or everything when the
        # overlap is off: gather what this pass produced; every rank issues the same collectives in the same order.  A
        # gradient left over from an EARLIER pass (the generator's, while the discriminator steps) is not touched.
        hooks = bool(self._foreign_hooks)
        for b in self._fbuckets:
            if b["work"] is not None:
                continue
            todo = [k for k, (p, _, _) in enumerate(b["members"])
                    if k not in b["have"] and p.grad is not None and (k in b["touched"] or not hooks)]
            if not todo and not b["have"]:
                continue                                     # nothing of this bucket took part in the pass (on any rank)
            for k in todo:
                self._foreign_stage(b, k)
            if b["work"] is None:                            # members without a gradient: their pieces travel as zeros
                buf = b["buf"]
                for k, (p, off, n) in enumerate(b["members"]):
                    if k not in b["have"]:
                        _cast(torch.zeros(n, dtype=buf.dtype, device=buf.device), buf[off:off + n])
                b["work"] = self._all_reduce_async(buf)
                self._count(b["numel"])
        for b in self._fbuckets:
            if b["work"] is None:
                continue
            b["work"].wait()
            if b["solo"]:
                p = b["members"][0][0]
                g = b.get("inplace")
                if g is None:
                    g = p.grad
                    if g.dtype == torch.float32 and g.is_contiguous():
                        _cast(b["buf"], g.view(-1))
                    else:
                        g.copy_(b["buf"].to(g.dtype).view_as(g))
                if g.dtype == torch.float32 and g.is_contiguous():
                    _scale(g.view(-1), scale)
                else:
                    g.mul_(scale)
            else:
                buf = b["buf"]
                if buf.dtype != torch.float32:
                    if self._f32_scratch is None or self._f32_scratch.numel() < buf.numel() or self._f32_scratch.device != buf.device:
                        self._f32_scratch = torch.empty(max(buf.numel(), self.bucket_elems + 8), dtype=torch.float32, device=buf.device)
                    wide = self._f32_scratch[:buf.numel()]
                    _cast(buf, wide)
                    buf = wide
                _scale(buf, scale)
                pairs = []
                for p, off, n in b["members"]:
                    g = p.grad
                    if g is None:
                        continue
                    if g.dtype == torch.float32 and g.is_contiguous():
                        pairs.append((buf[off:off + n], g.view(-1)))
                    else:
                        g.copy_(buf[off:off + n].to(g.dtype).view_as(g))
                _cast_many(pairs)
            b["work"], b["have"] = None, set()
            b.pop("inplace", None)
        for b in self._fbuckets:
            b["touched"] = set()

    def attach_engine(self, eng):
        eng.grad_ready_hook = self._on_ready
        self._engines.append(eng)

    # ---- overlapped path ----------------------------------------------------------------------
    def _state(self, arena):
        st = self._arenas.get(id(arena))
        if st is None:
            st = {"arena": arena, "sent": 0, "works": [], "active": False, "stage": None, "post": False}
            self._arenas[id(arena)] = st
        return st

    def subscribe(self, arena, fn) -> bool:
        """``fn(arena, lo, hi)`` is called for every bucket of the COMING backward pass of ``arena`` as soon as it is
        reduced and averaged -- on a stream of this object that waits for the collective, i.e. beside the rest of the
        backward pass (``optim.ArenaAdam.arm_streaming`` hangs the optimizer update on it: neither the x 1/R pass nor
        the 28 B/parameter Adam pass is left for the end of the step).  One-shot: ``finish()`` drops it.  HIP arenas
        only."""
        if not arena.flat.is_cuda or self.world < 2:
            return False
        self._subs[id(arena)] = fn
        return True

    def unsubscribe(self, arena) -> None:
        """Drop a subscription that has not fired (the subscriber gave up: ``ArenaAdam._disarm``)."""
        self._subs.pop(id(arena), None)

    def _launch(self, st, lo, hi):
        if hi <= lo:
            return
        arena = st["arena"]
        buf = arena.flat[lo:hi]
        if self.world > 1:
            stage = None
            if self.grad_dtype == torch.float32:
                work = self._all_reduce_async(buf)
            else:
                if st["stage"] is None:
                    st["stage"] = torch.empty(arena.flat.numel(), dtype=self.grad_dtype, device=buf.device)
                stage = st["stage"][lo:hi]
                _cast(buf, stage)                                  # fp32 -> bf16, ordered before the collective
                work = self._all_reduce_async(stage)
            fn = self._subs.get(id(arena))
            if fn is None:
                st["works"].append((work, stage, lo, hi))
            else:
                if self._post_stream is None:
                    self._post_stream = torch.cuda.Stream(device=buf.device)
                with torch.cuda.stream(self._post_stream):
                    work.wait()                                    # this stream (only) runs behind the collective
                    if stage is not None:
                        _cast(stage, buf)
                    _scale(buf, 1.0 / self.world)
                    fn(arena, lo, hi)
                st["post"] = True
                self.stats["post_buckets"] = self.stats.get("post_buckets", 0) + 1
        self.stats["buckets"] += 1
        self.stats["bytes"] += (hi - lo) * (4 if self.grad_dtype == torch.float32 else 2)

    def _all_reduce_async(self, buf):
        """In-place SUM all-reduce of ``buf``, ordered after the work already issued on the current stream; returns a
        handle whose wait() orders the current stream behind the collective."""
        if self.comm is None:
            return _TorchWork(buf, self.pg) if buf.is_cuda else \
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        from . import ops
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        ops.stream_wait(self._comm_stream, torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            self.comm.all_reduce(buf)
            if self._ev_next == len(self._events):
                self._events.append(ops.Event())
            ev = self._events[self._ev_next]      # pooled: the same events in the same order every step (finish() rewinds)
            self._ev_next += 1
            ev.record(self._comm_stream)
        buf.record_stream(self._comm_stream)
        return _CommWork(ev)

    def _on_ready(self, arena, end: int):
        """Engine callback: gradients in arena.flat[0:end] are final for this backward pass."""
        st = self._state(arena)
        if end < st["sent"]:
            raise RuntimeError("GradReducer: a second backward pass touched an arena whose buckets were "
                               "already reduced; use one backward per optimizer step")
        st["active"] = True
        if not self.overlap:
            return
        while end - st["sent"] >= self.bucket_elems:
            self._launch(st, st["sent"], st["sent"] + self.bucket_elems)
            st["sent"] += self.bucket_elems

    # ---- completion ------------------------------------------------------------------------------
    def finish(self):
        """Reduce what is left, wait, average.  Called once after each backward pass."""
        for st in self._arenas.values():
            if not st["active"]:
                continue
            total = st["arena"].flat.numel()
            self._launch(st, st["sent"], total)
            for w, stage, lo, hi in st["works"]:
                w.wait()
                if stage is not None:
                    _cast(stage, st["arena"].flat[lo:hi])          # reduced bf16 sum back into the fp32 master gradient
            if st["post"]:
                # every bucket of this arena was averaged (and handed to the subscriber) on the post stream
                _stream_wait(torch.cuda.current_stream(st["arena"].flat.device), self._post_stream)
            elif self.world > 1:
                _scale(st["arena"].flat, 1.0 / self.world)
            self._subs.pop(id(st["arena"]), None)
            st["sent"], st["works"], st["active"], st["post"] = 0, [], False, False
        if self._fbuckets and self.world > 1:
            self._finish_foreign()
        # rewound only when every handle of this step has been waited for: buckets _finish_foreign launches draw FRESH
        # events (re-recording one a hook-launched bucket still holds made its wait resolve later than needed), and the
        # next step starts at event 0 again -- the same events in the same order every step, which recorded plans rely on
        self._ev_next = 0


def broadcast_parameters(model: torch.nn.Module, src: int = 0, process_group=None):
    """Make every rank start from rank ``src``'s parameters and buffers (what DDP does at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if t.is_contiguous():
                dist.broadcast(t, src=src, group=process_group)
            else:
                # parameters of the HIP networks are dense but permuted ("fwd pack"): ship them in
                # storage order through a flat alias of the same memory
                flat = t.detach().as_strided((t.numel(),), (1,), t.storage_offset())
                dist.broadcast(flat, src=src, group=process_group)


def init_from_env(backend: Optional[str] = None):
    """torchrun-style environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("PAI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local % max(torch.cuda.device_count(), 1)))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
