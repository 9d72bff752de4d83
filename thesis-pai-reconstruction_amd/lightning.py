"""The slice of the pytorch_lightning 2.0 API that the reference's hot path is written
against, re-implemented without Lightning (which is not a dependency here).

Covers exactly what reference main.py:113-136, models/wrapper.py:117-173 and report.py:26-27
use: ``LightningModule`` manual-optimisation hooks (``optimizers``, ``toggle_optimizer``,
``untoggle_optimizer``, ``manual_backward``, ``log``, ``save_hyperparameters``, ``freeze``,
``load_from_checkpoint``), ``LightningDataModule``, ``Trainer.fit``, ``CSVLogger`` and
``ModelCheckpoint(monitor=..., mode=..., filename=...)``.  The ``Trainer`` runs one process per
GPU; gradient averaging across ranks is done by ``dist.GradReducer`` (RCCL all-reduce of the
flat gradient arenas, overlapped with the backward pass).
"""
from __future__ import annotations

import csv
import inspect
import os
import time
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------
class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.automatic_optimization = True
        self.trainer: Optional["Trainer"] = None
        self._pai_optimizers = None
        self._pai_toggle_state: Dict[Any, bool] = {}
        self._pai_hparams: Dict[str, Any] = {}
        self._pai_opt_steps = 0          # optimizer.step() calls: what Lightning 2.0 calls global_step under
        self.logged: Dict[str, Any] = {}  # manual optimisation (two per batch for loss_type="gan")

    # ---- hyper-parameters / checkpoints ------------------------------------------------
    def save_hyperparameters(self, *args, **kwargs):
        """Capture the constructor arguments of the calling ``__init__`` (Lightning semantics)."""
        frame = inspect.currentframe().f_back
        try:
            info = inspect.getargvalues(frame)
            hp = {}
            for name in info.args:
                if name == "self":
                    continue
                hp[name] = info.locals[name]
            if info.keywords:
                hp.update(info.locals.get(info.keywords, {}))
            self._pai_hparams = hp
        finally:
            del frame

    @property
    def hparams(self):
        return dict(self._pai_hparams)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
        ckpt = torch.load(str(checkpoint_path), map_location=map_location or "cpu", weights_only=False)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        if map_location is not None:
            model.to(map_location)
        return model

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad = True
        self.train()

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    # ---- logging ----------------------------------------------------------------------------
    def log(self, name, value, prog_bar=False, **kwargs):
        if self.trainer is not None:
            self.trainer._log(name, value)
        else:
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

    # ---- manual optimisation ---------------------------------------------------------------
    def configure_optimizers(self):
        raise NotImplementedError

    def optimizers(self):
        if self._pai_optimizers is None:
            o = self.configure_optimizers()
            self._pai_optimizers = list(o) if isinstance(o, (tuple, list)) else o
            for opt in (self._pai_optimizers if isinstance(self._pai_optimizers, list) else [self._pai_optimizers]):
                self._count_steps(opt)
        return self._pai_optimizers

    def _count_steps(self, opt):
        """Lightning wraps every optimizer so that each ``step()`` advances ``trainer.global_step``; with manual
        optimisation and two optimizers (reference models/wrapper.py:136,160) that is two per batch, which is what
        ``--steps`` (max_steps, reference main.py:125) and the checkpoint's ``global_step`` count."""
        if getattr(opt, "_pai_counted", False):
            return
        orig = opt.step

        def step(*args, **kwargs):
            out = orig(*args, **kwargs)
            self._pai_opt_steps += 1
            return out
        opt.step = step
        opt._pai_counted = True

    def _all_optimizers(self) -> List[torch.optim.Optimizer]:
        o = self.optimizers()
        return o if isinstance(o, list) else [o]

    def toggle_optimizer(self, optimizer):
        """Lightning 2.0: remember requires_grad of every parameter of every optimizer, switch
        them all off, then restore the ones owned by ``optimizer``."""
        state = {}
        for opt in self._all_optimizers():
            for group in opt.param_groups:
                for p in group["params"]:
                    if p in state:
                        continue
                    state[p] = p.requires_grad
                    p.requires_grad = False
        for group in optimizer.param_groups:
            for p in group["params"]:
                p.requires_grad = state[p]
        self._pai_toggle_state = state

    def untoggle_optimizer(self, optimizer):
        for opt in self._all_optimizers():
            if opt is optimizer:
                continue
            for group in opt.param_groups:
                for p in group["params"]:
                    if p in self._pai_toggle_state:
                        p.requires_grad = self._pai_toggle_state[p]
        self._pai_toggle_state = {}

    def manual_backward(self, loss, *args, **kwargs):
        if not args and not kwargs and loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
            # the implicit seed of a scalar backward is a fresh ones_like (a fill launch per call); a cached one is
            # recognised by the fused loss functions, which then skip their `grad * seed` launches as well
            from . import functional as PF
            loss.backward(gradient=PF.unit_seed(loss.device))
        else:
            loss.backward(*args, **kwargs)
        from . import nnops
        nnops.join_wgrads()      # composable networks: weight gradients run on a second stream (nnops._WgradStream)
        if self.trainer is not None and self.trainer.reducer is not None:
            self.trainer.reducer.finish()

    # ---- default hooks -----------------------------------------------------------------------
    def set_precision(self, precision: str):
        """Called by the Trainer with the ``--precision`` string (reference main.py:129)."""
        dtype = precision_to_dtype(precision)
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype


def precision_to_dtype(precision) -> torch.dtype:
    p = str(precision)
    if p in ("32", "32-true", "64", "64-true"):
        return torch.float32
    if p in ("bf16", "bf16-mixed", "bf16-true"):
        return torch.bfloat16
    if p in ("16", "16-mixed", "16-true"):
        # fp16 autocast (loss scaling, 5 exponent bits) has no counterpart on the HIP path: refuse rather than
        # silently train in another format
        raise ValueError(f"precision {precision!r} (fp16) is not built; use 'bf16-mixed' (bf16 storage, fp32 "
                         f"accumulate) or '32'")
    raise ValueError(f"unknown precision {precision!r}")


class LightningDataModule:
    def __init__(self):
        pass

    def setup(self, stage: str):
        pass


class Callback:
    def on_fit_start(self, trainer, pl_module): pass
    def on_train_batch_end(self, trainer, pl_module, *a, **k): pass
    def on_validation_start(self, trainer, pl_module): pass
    def on_validation_end(self, trainer, pl_module): pass


# --------------------------------------------------------------------------------------
class CSVLogger:
    """``pl.loggers.CSVLogger(save_dir, name)`` -> save_dir/name/version_k/metrics.csv."""

    def __init__(self, save_dir, name="lightning_logs", version=None):
        self.save_dir, self.name = str(save_dir), name
        root = os.path.join(self.save_dir, name)
        if version is None:
            version = 0
            if os.path.isdir(root):
                taken = [int(d.split("_")[1]) for d in os.listdir(root)
                         if d.startswith("version_") and d.split("_")[1].isdigit()]
                version = max(taken) + 1 if taken else 0
        self.version = version
        self.log_dir = os.path.join(root, f"version_{version}")
        self.rows: List[Dict[str, Any]] = []
        self._keys: List[str] = []

    def log_metrics(self, metrics: Dict[str, float], step: int):
        row = dict(metrics)
        row["step"] = step
        for k in row:
            if k not in self._keys:
                self._keys.append(k)
        self.rows.append(row)

    def save(self):
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "metrics.csv"), "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=self._keys)
            wr.writeheader()
            wr.writerows(self.rows)


class ModelCheckpoint(Callback):
    """Keeps the checkpoint with the best monitored validation metric
    (reference main.py:113-119: monitor="val_ssim", mode="max", filename="best")."""

    def __init__(self, save_top_k=1, monitor=None, mode="min", filename=None, save_last=False, dirpath=None):
        self.monitor, self.mode = monitor, mode
        self.filename = filename or "checkpoint"
        self.save_last = bool(save_last)
        self.dirpath = dirpath
        self.best_model_score = None
        self.best_model_path = ""

    def state_dict(self):
        """Carried in checkpoints (Trainer.save_checkpoint) so that a resumed run does not overwrite best.ckpt with a
        worse first validation."""
        score = self.best_model_score
        return {"monitor": self.monitor, "best_model_score": None if score is None else float(score),
                "best_model_path": self.best_model_path}

    def load_state_dict(self, state):
        if state.get("monitor") == self.monitor:
            self.best_model_score = state.get("best_model_score")
            self.best_model_path = state.get("best_model_path", "")

    def _dir(self, trainer):
        if self.dirpath:
            return self.dirpath
        base = trainer.logger.log_dir if trainer.logger is not None else trainer.default_root_dir
        return os.path.join(base, "checkpoints")

    def on_validation_end(self, trainer, pl_module):
        if trainer.global_rank != 0:
            return
        score = trainer.callback_metrics.get(self.monitor) if self.monitor else None
        better = True
        if self.monitor is not None:
            if score is None:
                return
            if self.best_model_score is not None:
                better = score > self.best_model_score if self.mode == "max" else score < self.best_model_score
        d = self._dir(trainer)
        os.makedirs(d, exist_ok=True)
        if better:
            self.best_model_score = score
            self.best_model_path = os.path.join(d, self.filename + ".ckpt")
            trainer.save_checkpoint(self.best_model_path)
        if self.save_last:
            trainer.save_checkpoint(os.path.join(d, "last.ckpt"))


class DevicePrefetcher:
    """Iterates a loader one batch AHEAD of the consumer: while step i runs, batch i + 1 is already being copied host ->
    device on a copy stream of its own (pinned staging, non_blocking), so the H2D transfer of 2 x N x 256 x 256 fp32
    never sits on the step's critical path (reference: Lightning moves each batch synchronously, main.py:136).
    On a CPU "device" (tests) it is a plain pass-through."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        dev = self.device
        if dev is None or torch.device(dev).type != "cuda":
            for batch in self.loader:
                yield batch if dev is None else tuple(b.to(dev) if torch.is_tensor(b) else b for b in batch)
            return
        copy_stream = torch.cuda.Stream(dev)
        it = iter(self.loader)

        def stage():
            batch = next(it, None)
            if batch is None:
                return None
            with torch.cuda.stream(copy_stream):
                out = tuple((b if b.is_pinned() else b.pin_memory()).to(dev, non_blocking=True)
                            if torch.is_tensor(b) else b for b in batch)
            done = torch.cuda.Event()
            done.record(copy_stream)
            return out, done

        nxt = stage()
        while nxt is not None:
            cur, done = nxt
            main = torch.cuda.current_stream(dev)
            main.wait_event(done)
            for b in cur:
                if torch.is_tensor(b):
                    b.record_stream(main)        # allocated on the copy stream, consumed on the main one
            nxt = stage()                        # the next copy is in flight while the caller trains on `cur`
            yield cur


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu().contiguous()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


# --------------------------------------------------------------------------------------
class Trainer:
    def __init__(self, max_epochs=None, max_steps=-1, log_every_n_steps=50, check_val_every_n_epoch=1,
                 logger=None, precision="32", callbacks=None, benchmark=None, default_root_dir=None,
                 enable_progress_bar=True, device=None, reducer=None, **unused):
        self.max_epochs = max_epochs if max_epochs is not None else (1000 if max_steps == -1 else None)
        self.max_steps = max_steps
        self.log_every_n_steps = log_every_n_steps
        self.check_val_every_n_epoch = check_val_every_n_epoch
        lg = logger[0] if isinstance(logger, (list, tuple)) and logger else logger
        self.logger: Optional[CSVLogger] = lg
        self.precision = str(precision)
        self.callbacks: List[Callback] = list(callbacks or [])
        self.default_root_dir = default_root_dir or os.getcwd()
        self.enable_progress_bar = enable_progress_bar
        self.device = device
        self.reducer = reducer
        self.global_step = 0        # optimizer steps (Lightning 2.0 semantics): max_steps, checkpoint field
        self.batches_seen = 0       # training batches: logging cadence and the CSV "step" column
        self.current_epoch = 0
        self.callback_metrics: Dict[str, float] = {}
        self._step_logs: Dict[str, Any] = {}
        self._val_acc: Optional[Dict[str, list]] = None
        self._val_bs = 1
        self.model: Optional[LightningModule] = None
        import torch.distributed as dist
        self.global_rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    # ---- logging plumbing -----------------------------------------------------------------------
    def _log(self, name, value):
        v = value.detach() if torch.is_tensor(value) else value
        if self._val_acc is not None:
            self._val_acc.setdefault(name, []).append((v, self._val_bs))
        else:
            self._step_logs[name] = v

    def _flush_step_logs(self):
        if not self._step_logs:
            return
        vals = {k: float(v) for k, v in self._step_logs.items()}   # host sync, every n steps only
        self.callback_metrics.update(vals)
        if self.logger is not None and self.global_rank == 0:
            vals["epoch"] = self.current_epoch
            self.logger.log_metrics(vals, self.batches_seen)

    def save_checkpoint(self, path):
        """Lightning-2.0-shaped checkpoint: weights, hyper-parameters, optimizer states (one per configured optimizer,
        in ``configure_optimizers`` order) and the state of every callback that has one (the EMA shadow weights:
        the reference returns them from ``on_save_checkpoint``, callbacks/ema.py:54-62) -- enough to resume."""
        m = self.model
        opts = m._all_optimizers() if m._pai_optimizers is not None else []
        ckpt = {
            "epoch": self.current_epoch,
            "global_step": self.global_step,
            "batches_seen": self.batches_seen,
            "pytorch-lightning_version": "2.0.2",
            "state_dict": {k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()},
            "hyper_parameters": m.hparams,
            "optimizer_states": [_to_cpu(o.state_dict()) for o in opts],
            "callbacks": {type(cb).__name__: cb.state_dict() for cb in self.callbacks if hasattr(cb, "state_dict")},
        }
        tmp = str(path) + ".tmp"
        torch.save(ckpt, tmp)
        os.replace(tmp, str(path))

    def restore(self, model: "LightningModule", checkpoint_path):
        """Load weights, optimizer states, callback states and the step counters saved by ``save_checkpoint``;
        call before ``fit`` (``fit(..., ckpt_path=...)`` does)."""
        ckpt = torch.load(str(checkpoint_path), map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["state_dict"])
        if self.device is not None:
            model.to(self.device)
        opts = model._all_optimizers()
        for o, st in zip(opts, ckpt.get("optimizer_states", [])):
            o.load_state_dict(st)
        for cb in self.callbacks:
            st = ckpt.get("callbacks", {}).get(type(cb).__name__)
            if st is not None and hasattr(cb, "load_state_dict"):
                cb.load_state_dict(st)
        self.global_step = int(ckpt.get("global_step", 0))
        model._pai_opt_steps = self.global_step
        self.batches_seen = int(ckpt.get("batches_seen", 0))
        # checkpoints are written by ModelCheckpoint after the validation that closes an epoch: continue with the next
        self.current_epoch = int(ckpt.get("epoch", -1)) + 1
        return ckpt

    # ---- loops -------------------------------------------------------------------------------------
    def _to_device(self, batch):
        dev = self.device
        if dev is None:
            return batch
        return tuple(b.to(dev, non_blocking=True) if torch.is_tensor(b) else b for b in batch)

    def validate_epoch(self, model, loader):
        model.eval()
        for cb in self.callbacks:
            cb.on_validation_start(self, model)
        self._val_acc = {}
        with torch.no_grad():
            for bi, batch in enumerate(DevicePrefetcher(loader, self.device)):
                self._val_bs = int(batch[0].shape[0])
                model.validation_step(batch, bi)
        acc, self._val_acc = self._val_acc, None
        out = {}
        keys = sorted(acc)
        sums = torch.tensor([[sum(float(v) * bs for v, bs in acc[k]), float(sum(bs for _, bs in acc[k]))] for k in keys],
                            dtype=torch.float64).reshape(len(keys), 2)
        if self.world_size > 1 and keys:
            # the validation split is sharded across the ranks: batch-size-weighted mean over ALL ranks, so that
            # every rank (and the ModelCheckpoint on rank 0) sees the same epoch value
            import torch.distributed as dist
            buf = sums.to(self.device) if (self.device is not None and dist.get_backend() == "nccl") else sums
            dist.all_reduce(buf)
            sums = buf.cpu()
        for i, k in enumerate(keys):
            out[k] = float(sums[i, 0]) / max(float(sums[i, 1]), 1.0)
        self.callback_metrics.update(out)
        if self.logger is not None and self.global_rank == 0 and out:
            row = dict(out)
            row["epoch"] = self.current_epoch
            self.logger.log_metrics(row, self.batches_seen)
            self.logger.save()
        for cb in self.callbacks:
            cb.on_validation_end(self, model)
        model.train()
        return out

    def fit(self, model: LightningModule, datamodule=None, train_dataloaders=None, val_dataloaders=None,
            ckpt_path=None):
        self.model = model
        model.trainer = self
        if self.device is not None:
            model.to(self.device)
        model.set_precision(self.precision)
        if datamodule is not None:
            datamodule.setup("fit")
            train_loader = datamodule.train_dataloader()
            val_loader = datamodule.val_dataloader() if hasattr(datamodule, "val_dataloader") else None
        else:
            train_loader, val_loader = train_dataloaders, val_dataloaders
        model.train()
        model.optimizers()
        if self.reducer is not None:
            self.reducer.attach(model)
        for cb in self.callbacks:
            cb.on_fit_start(self, model)
        if ckpt_path is not None:
            self.restore(model, ckpt_path)
        # The training step as one C call (plan.PlannedStep): recorded after a few eager steps, replayed from then on;
        # anything it cannot own (torch-launched kernels inside the step, dropout masks, CPU tensors) makes it step aside
        # and the call below IS model.training_step.  PAI_PLAN=0 turns it off.
        step_fn = model.training_step
        self.planned_step = None
        if self.device is not None and torch.device(self.device).type == "cuda":
            from . import plan as _plan
            if _plan.enabled_by_default():
                self.planned_step = step_fn = _plan.PlannedStep(model)
        t0 = time.time()
        done = False
        epoch = self.current_epoch
        while not done and (self.max_epochs is None or epoch < self.max_epochs):
            self.current_epoch = epoch
            if hasattr(train_loader, "set_epoch"):
                train_loader.set_epoch(epoch)
            for bi, batch in enumerate(DevicePrefetcher(train_loader, self.device)):
                self._step_logs = {}
                step_fn(batch, bi)
                self.batches_seen += 1
                self.global_step = model._pai_opt_steps
                for cb in self.callbacks:
                    cb.on_train_batch_end(self, model, None, batch, bi)
                if self.batches_seen % self.log_every_n_steps == 0:
                    self._flush_step_logs()
                    if self.enable_progress_bar and self.global_rank == 0:
                        msg = " ".join(f"{k}={v:.4g}" for k, v in self.callback_metrics.items()
                                       if not k.startswith("val_"))
                        print(f"[epoch {epoch} batch {self.batches_seen} step {self.global_step} "
                              f"{time.time() - t0:.1f}s] {msg}", flush=True)
                if self.max_steps != -1 and self.global_step >= self.max_steps:
                    done = True
                    break
            if val_loader is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                out = self.validate_epoch(model, val_loader)
                if self.enable_progress_bar and self.global_rank == 0:
                    print(f"[epoch {epoch} validation] " + " ".join(f"{k}={v:.5g}" for k, v in out.items()),
                          flush=True)
            epoch += 1
        if self.logger is not None and self.global_rank == 0:
            self.logger.save()
        model.trainer = None
        return model
