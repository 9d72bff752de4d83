"""Callbacks of the reference's training CLI.

``EMACallback`` mirrors reference callbacks/ema.py:5-72, which wraps
``torch_ema.ExponentialMovingAverage`` (torch_ema 0.3: shadow = shadow - (1 - d_t) * (shadow - p)
with the warm-up decay d_t = min(decay, (1 + n) / (10 + n)) after n updates) over ALL module
parameters, updates it after every training batch and swaps it in for validation.
On the HIP device the update is ONE C-ABI launch per 48 memory segments (``pai_lerp_multi``): the shadow copies live
in one flat buffer laid out in the ADDRESS order of the parameters, and neighbours in memory are merged into one
segment -- the parameters of a HIP engine sit back to back in its arena (``engine.GradArena.pflat``), so a network is
one segment and the whole update a single fused pass (the composable networks' separately allocated parameters: a few
launches of 48 tensors, like ``optim.MultiAdam``).  Host tensors take ``torch._foreach`` ops of the same arithmetic.
"""
from __future__ import annotations

from typing import List

import torch

from .lightning import Callback


class EMACallback(Callback):
    def __init__(self, decay: float = 0.9999, use_num_updates: bool = True):
        self.decay = decay
        self.use_num_updates = use_num_updates
        self.num_updates = 0
        self.shadow: List[torch.Tensor] = []
        self.backup: List[torch.Tensor] = []
        self.params: List[torch.nn.Parameter] = []

    def on_fit_start(self, trainer, pl_module):
        self.params = [p for p in pl_module.parameters() if p.requires_grad]
        self.shadow = [p.detach().clone(memory_format=torch.preserve_format) for p in self.params]
        self.num_updates = 0
        self._segments, self._ptrs, self._flat = None, None, None

    # ---- fused device update ------------------------------------------------------------------------------
    @staticmethod
    def _dense(t) -> bool:
        """Non-overlapping and dense: some permutation of a contiguous block (the fwd-pack views of the HIP engines)."""
        if t.is_contiguous():
            return True
        expect = 1
        for stride, size in sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz != 1):
            if stride != expect:
                return False
            expect *= size
        return True

    def _fusable(self) -> bool:
        return bool(self.params) and all(p.is_cuda and p.dtype == torch.float32 and self._dense(p) for p in self.params)

    def _build_segments(self):
        """Shadow storage = one flat fp32 buffer in the address order of the parameters (values carried over); adjacent
        (parameter, shadow) pairs become one segment.  Rebuilt whenever a parameter has moved -- the first optimizer step
        of a HIP engine moves its parameters into the arena."""
        order = sorted(range(len(self.params)), key=lambda i: self.params[i].data_ptr())
        total = sum(self.params[i].numel() for i in order)
        flat = torch.empty(total, dtype=torch.float32, device=self.params[0].device)
        segs, off = [], 0
        for i in order:
            p, old = self.params[i], self.shadow[i]
            n = p.numel()
            view = flat[off:off + n].as_strided(p.shape, p.stride())        # same (dense, possibly permuted) layout as p
            view.copy_(old)
            self.shadow[i] = view
            dptr, sptr = flat.data_ptr() + 4 * off, p.data_ptr()
            if segs and segs[-1][0] + 4 * segs[-1][2] == dptr and segs[-1][1] + 4 * segs[-1][2] == sptr:
                segs[-1] = (segs[-1][0], segs[-1][1], segs[-1][2] + n)
            else:
                segs.append((dptr, sptr, n))
            off += n
        self._flat, self._segments = flat, segs
        self._ptrs = [p.data_ptr() for p in self.params]

    def _decay(self) -> float:
        if not self.use_num_updates:
            return self.decay
        return min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))

    @torch.no_grad()
    def on_train_batch_end(self, trainer, pl_module, *args, **kwargs):
        self.num_updates += 1
        one_minus = 1.0 - self._decay()
        if self._fusable():
            from . import ops
            if self._segments is None or self._ptrs != [p.data_ptr() for p in self.params]:
                self._build_segments()
            ops.lerp_multi(self._segments, one_minus)
            return
        # torch_ema's arithmetic: shadow -= (1 - d) * (shadow - p)
        tmp = torch._foreach_sub(self.shadow, [p.detach() for p in self.params])
        torch._foreach_mul_(tmp, one_minus)
        torch._foreach_sub_(self.shadow, tmp)

    @torch.no_grad()
    def on_validation_start(self, trainer, pl_module):
        self.backup = [p.detach().clone(memory_format=torch.preserve_format) for p in self.params]
        for p, s in zip(self.params, self.shadow):
            p.copy_(s)
        _bump_weight_generation(pl_module)

    @torch.no_grad()
    def on_validation_end(self, trainer, pl_module):
        for p, b in zip(self.params, self.backup):
            p.copy_(b)
        self.backup = []
        _bump_weight_generation(pl_module)

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates,
                "shadow_params": [s.detach().cpu().contiguous() for s in self.shadow]}

    def load_state_dict(self, state):
        """Counterpart of ``state_dict`` (torch_ema's load_state_dict, reference callbacks/ema.py:64-72): call after
        ``on_fit_start`` has sized the shadow list."""
        self.decay = state["decay"]
        self.num_updates = int(state["num_updates"])
        shadow = state["shadow_params"]
        if len(shadow) != len(self.shadow):
            raise ValueError(f"EMA state holds {len(shadow)} tensors, the module has {len(self.shadow)} parameters")
        with torch.no_grad():
            for dst, src in zip(self.shadow, shadow):
                dst.copy_(src.to(dst.device).reshape(dst.shape))


def _bump_weight_generation(module: torch.nn.Module):
    """copy_() bumps torch's version counters, which the engines watch; nothing else to do -- kept as
    an explicit hook so that raw-pointer writers have one place to announce weight changes."""
    for m in module.modules():
        eng = getattr(m, "_engine", None)
        if eng is not None and hasattr(eng, "weights_generation"):
            eng.weights_generation[0] += 1
