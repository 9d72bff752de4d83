"""Minimal stand-in for pytorch_lightning 2.0.2, used ONLY by oracle/gen_golden.py
to import the real reference model classes in the build container (the real
package is not installed and there is no network).  Reproduces the
manual-optimisation hooks the reference's training_step uses
(models/wrapper.py:121-162): optimizers(), toggle_optimizer(),
untoggle_optimizer(), manual_backward(), log().  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn as nn
from . import callbacks, loggers  # noqa: F401


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.automatic_optimization = True
        self._opts = None
        self._toggle_state = {}
        self.logged = {}

    # -- hooks used by the reference -------------------------------------
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, name, value, **kw):
        self.logged[name] = value.detach().clone() if torch.is_tensor(value) else value

    @property
    def device(self):
        return next(self.parameters()).device

    def optimizers(self):
        if self._opts is None:
            o = self.configure_optimizers()
            self._opts = list(o) if isinstance(o, (tuple, list)) else o
        return self._opts

    def toggle_optimizer(self, optimizer):
        # pl 2.0.2: remember requires_grad of every param of every optimizer,
        # switch all off, then restore the active optimizer's own params.
        opts = self.optimizers()
        opts = opts if isinstance(opts, list) else [opts]
        state = {}
        for opt in opts:
            for group in opt.param_groups:
                for p in group["params"]:
                    if p in state:
                        continue
                    state[p] = p.requires_grad
                    p.requires_grad = False
        for group in optimizer.param_groups:
            for p in group["params"]:
                p.requires_grad = state[p]
        self._toggle_state = state

    def untoggle_optimizer(self, optimizer):
        opts = self.optimizers()
        opts = opts if isinstance(opts, list) else [opts]
        for opt in opts:
            if opt is optimizer:
                continue
            for group in opt.param_groups:
                for p in group["params"]:
                    if p in self._toggle_state:
                        p.requires_grad = self._toggle_state[p]
        self._toggle_state = {}

    def manual_backward(self, loss):
        loss.backward()

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()


class LightningDataModule:
    def __init__(self):
        pass


class Trainer:  # placeholder so `pl.Trainer` resolves at import time
    def __init__(self, *a, **k):
        raise RuntimeError("shim")
