"""Callbacks of the reference's training CLI.

``EMACallback`` mirrors reference callbacks/ema.py:5-72, which wraps
``torch_ema.ExponentialMovingAverage`` (torch_ema 0.3: shadow = shadow - (1 - d_t) * (shadow - p)
with the warm-up decay d_t = min(decay, (1 + n) / (10 + n)) after n updates) over ALL module
parameters, updates it after every training batch and swaps it in for validation.
The update is a multi-tensor lerp; when the parameters live in the flat arenas of the HIP engines it
is a single fused pass per network.
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

    def _decay(self) -> float:
        if not self.use_num_updates:
            return self.decay
        return min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))

    @torch.no_grad()
    def on_train_batch_end(self, trainer, pl_module, *args, **kwargs):
        self.num_updates += 1
        one_minus = 1.0 - self._decay()
        # shadow += (1 - d) * (p - shadow)
        torch._foreach_lerp_(self.shadow, [p.detach() for p in self.params], one_minus)

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
