"""Layer executors: run the reference's generator and PatchGAN topologies on the HIP kernels.

``UnetEngine`` restates the data flow of ``Unet.forward`` (reference models/pix2pix.py:198-216)
and its autograd backward as an explicit schedule of C-ABI calls; ``DiscEngine`` does the same
for ``Discriminator.forward`` (reference models/wrapper.py:236-238).  Nothing is computed in
Python: the engines only own buffers (torch tensors as device memory), filter packs, the flat
gradient arena and the launch order.

Data layout in HBM
  * activations NHWC in the compute dtype (fp32 or bf16); for every BatchNorm layer both the raw
    convolution output ``z`` (needed by the BN backward) and the normalised+activated tensor
    ``a`` (what the consumers read) are kept;
  * ``torch.cat`` is never materialised: consumers read two tensors, input gradients are
    written to two tensors (reference models/pix2pix.py:212);
  * parameters stay fp32 ``nn.Parameter``s with the reference's logical shapes and state-dict
    keys, but are physically stored in the kernels' "fwd pack" [Cout][kh][kw][Cin]; gradients
    are views into one flat fp32 arena laid out in backward-completion order so that
    data-parallel buckets are contiguous ranges that become ready front to back.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH

BN_EPS_DEFAULT = 1e-5


# --------------------------------------------------------------------------------------
# parameter layout helpers
# --------------------------------------------------------------------------------------
def _fwd_pack_strides(mod) -> tuple:
    if isinstance(mod, nn.ConvTranspose2d):
        cin, cout, kh, kw = mod.weight.shape
        return (1, kh * kw * cin, kw * cin, cin)
    cout, cin, kh, kw = mod.weight.shape
    return (kh * kw * cin, 1, kw * cin, cin)


def _strides_match(t: torch.Tensor, want: tuple) -> bool:
    return all(s == w or n == 1 for s, w, n in zip(t.stride(), want, t.shape))


def to_fwd_pack_(mod) -> None:
    """Re-stride ``mod.weight`` in place to physical [Cout][kh][kw][Cin] (logical shape unchanged)."""
    w = mod.weight
    if _strides_match(w, _fwd_pack_strides(mod)) and w.permute(_phys_perm(mod)).is_contiguous():
        return
    with torch.no_grad():
        w.data = w.data.permute(_phys_perm(mod)).contiguous().permute(_logical_perm(mod))


def _phys_perm(mod):
    return (1, 2, 3, 0) if isinstance(mod, nn.ConvTranspose2d) else (0, 2, 3, 1)


def _logical_perm(mod):
    return (3, 0, 1, 2) if isinstance(mod, nn.ConvTranspose2d) else (0, 3, 1, 2)


def _cin_cout(mod):
    if isinstance(mod, nn.ConvTranspose2d):
        return mod.weight.shape[0], mod.weight.shape[1]
    return mod.weight.shape[1], mod.weight.shape[0]


class GradArena:
    """One flat fp32 buffer holding every parameter gradient of a network, ordered by the
    point in the backward pass at which it becomes final."""

    ALIGN = 64

    def __init__(self, entries, device):
        # entries: list of (param, conv_module_or_None)
        self.offsets = {}
        off = 0
        for p, _ in entries:
            self.offsets[id(p)] = (off, p.numel())
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.views = {}
        self.params = []
        self._zero_small = self._zero_all = None
        self._small = []      # segments that are ACCUMULATED into (BatchNorm gamma / beta) or never written (conv biases in
        #                       front of a BatchNorm): cleared per backward pass; the conv weights are overwritten instead
        for p, mod in entries:
            o, n = self.offsets[id(p)]
            seg = self.flat[o:o + n]
            if mod is not None and p.dim() == 4:
                cin, cout = _cin_cout(mod)
                v = seg.view(cout, 4, 4, cin).permute(_logical_perm(mod))
                if min(cin, cout) <= 2:       # thin layers: their kernels ADD partial tiles -- cleared with the small segments
                    self._small.append(seg)
            else:
                v = seg.view(p.shape)
                self._small.append(seg)
            self.views[id(p)] = v
            self.params.append(p)

    def view(self, p):
        return self.views[id(p)]

    def seg(self, p):
        o, n = self.offsets[id(p)]
        return self.flat[o:o + n]

    def end_of(self, p) -> int:
        o, n = self.offsets[id(p)]
        return (o + n + self.ALIGN - 1) // self.ALIGN * self.ALIGN

    def begin_backward(self, params, overwrite_weights: bool = False) -> bool:
        """Prepare the arena for a backward pass; returns True for the FIRST pass since ``zero_grad`` (``params`` carry
        no gradients yet), False when they already carry this arena's views (gradient accumulation across several
        backward passes: nothing is cleared).  First pass: the arena is zeroed -- or, with ``overwrite_weights``, only
        its small segments are (one multi-tensor launch instead of a 218 MB fill for the Pix2Pix generator): the caller
        then promises to OVERWRITE every conv-weight segment (``pai_conv_wgrad_overwrite``) in that pass."""
        grads = [p.grad for p in params if p.requires_grad]
        if all(g is None for g in grads):
            if overwrite_weights and self._small:
                if self.flat.is_cuda:
                    if self._zero_small is None:
                        self._zero_small = ops.ZeroList(self._small)
                    self._zero_small()       # one pai_zero_multi launch: a node of the launch plan like everything else
                else:
                    torch._foreach_zero_(self._small)
            elif self.flat.is_cuda:
                if self._zero_all is None:
                    self._zero_all = ops.ZeroList([self.flat])
                self._zero_all()
            else:
                self.flat.zero_()
            return True
        for p in params:
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.view(p).data_ptr()):
                raise ops.PaiError("gradient arena: parameters carry foreign .grad tensors; call "
                                   "zero_grad(set_to_none=True) before backward")
        return False

    def attach(self, params) -> None:
        for p in params:
            if p.requires_grad:
                p.grad = self.view(p)

    # ---- flat parameter / Adam-moment arenas (same layout as the gradient arena) ----------------
    def adopt_parameters(self) -> bool:
        """Move the parameters into one flat fp32 buffer laid out like the gradient arena (each
        ``p.data`` becomes a strided view of it) and allocate the two Adam moment buffers.  Returns
        False when the parameters are not on this arena's device."""
        dev = self.flat.device
        if any(p.device != dev for p in self.params):
            return False
        if getattr(self, "pflat", None) is not None and self.params_adopted():
            return True
        pflat = torch.zeros_like(self.flat)
        with torch.no_grad():
            for p in self.params:
                o, n = self.offsets[id(p)]
                v = self.views[id(p)]
                pv = pflat[o:o + n].as_strided(v.shape, v.stride(), o)
                pv.copy_(p.data)
                p.data = pv
        self.pflat = pflat
        if getattr(self, "mflat", None) is None:
            self.mflat = torch.zeros_like(self.flat)
            self.vflat = torch.zeros_like(self.flat)
        return True

    def params_adopted(self) -> bool:
        pf = getattr(self, "pflat", None)
        if pf is None:
            return False
        base = pf.data_ptr()
        return all(p.data_ptr() == base + 4 * self.offsets[id(p)][0] for p in self.params)

    def moment_views(self, p):
        o, n = self.offsets[id(p)]
        v = self.views[id(p)]
        return (self.mflat[o:o + n].as_strided(v.shape, v.stride(), o),
                self.vflat[o:o + n].as_strided(v.shape, v.stride(), o))


class _Packs:
    """Compute-dtype filter packs of one conv layer, refreshed when the master weight changes."""

    def __init__(self, mod, need_dgrad: bool, gen):
        self.mod = mod
        self.need_dgrad = need_dgrad
        # gen[0] is bumped by optimizers that update the master weights through raw pointers
        # (ArenaAdam), which torch's per-tensor version counter does not see
        self.gen = gen
        self.version = None
        self.dtype = None
        self.wf = None
        self.wd = None

    def _stale(self, dtype):
        w = self.mod.weight
        return self.version != (w._version, w.data_ptr(), self.gen[0]) or self.dtype != dtype

    def _prepare(self, dtype):
        """Buffers of a stale pack: (master, Cout, taps, Cin, fwd pack to write | None, dgrad pack to write | None)."""
        to_fwd_pack_(self.mod)
        w = self.mod.weight
        cin, cout = _cin_cout(self.mod)
        if dtype == torch.float32:
            self.wf = w  # the parameter storage IS the fp32 fwd pack
            wf_out = None
        else:
            if self.wf is None or self.wf.dtype != dtype or self.wf is w:
                self.wf = torch.empty(w.numel(), dtype=dtype, device=w.device)
            wf_out = self.wf
        wd_out = None
        if self.need_dgrad:
            if self.wd is None or self.wd.dtype != dtype:
                self.wd = torch.empty(w.numel(), dtype=dtype, device=w.device)
            wd_out = self.wd
        return w, cout, w.shape[2] * w.shape[3], cin, wf_out, wd_out

    def _mark(self, dtype):
        w = self.mod.weight
        self.version = (w._version, w.data_ptr(), self.gen[0])
        self.dtype = dtype

    # ---- packs written by the optimizer itself (pai_adam_pack) ---------------------------------------------------------
    def spare(self):
        """The second set of pack buffers, for an update that runs while the current packs are still being read (the
        input-gradient kernels of the SAME backward pass use the weights of before the update): None unless the layer
        has bf16 packs in use.  The optimizer writes them and ``commit``s once the step is complete."""
        if self.dtype != torch.bfloat16 or self.wf is None or self.wf is self.mod.weight:
            return None
        if getattr(self, "_spare", None) is None:
            self._spare = (torch.empty_like(self.wf), torch.empty_like(self.wd) if self.wd is not None else None)
        return self._spare

    def commit(self):
        """The spare buffers now hold the packs of the CURRENT master weights (call after the generation bump)."""
        (self.wf, self.wd), self._spare = self._spare, (self.wf, self.wd)
        self._mark(self.dtype)

    def commit_replayed(self):
        """``commit`` for a step replayed from a launch plan: the recorded Adam launch has filled the spare set."""
        self.commit()

    def get(self, dtype):
        if self._stale(dtype):
            w, cout, taps, cin, wf_out, wd_out = self._prepare(dtype)
            if wf_out is not None or wd_out is not None:
                ops.pack_weights(dtype, w, cout, taps, cin, wf_out, wd_out)
            self._mark(dtype)
        return self.wf, self.wd


def refresh_packs(packs, dtype) -> None:
    """Re-pack every stale layer of a network; the bf16 layers whose channel counts are multiples of 64 share ONE launch
    (after an optimizer step all of them are stale: 17 launches of ~8 us in the Pix2Pix step otherwise).  The rest is
    left to ``_Packs.get``."""
    if dtype != torch.bfloat16:
        return
    batch, todo = [], []
    for pk in packs:
        if not pk._stale(dtype):
            continue
        cin, cout = _cin_cout(pk.mod)
        if cin % 64 or cout % 64:
            continue
        item = pk._prepare(dtype)
        batch.append(item)
        todo.append(pk)
    if batch:
        ops.pack_weights_multi(batch)
        for pk in todo:
            pk._mark(dtype)


def pack_targets(arena, packs):
    """The layers whose packs an optimizer may write itself (``pai_adam_pack``): [(arena offset, numel, Cout, taps, Cin,
    _Packs)] sorted by offset -- dense bf16 layers with channel counts in multiples of 64, as ``refresh_packs`` batches."""
    out = []
    for pk in packs:
        cin, cout = _cin_cout(pk.mod)
        if cin % 64 or cout % 64 or id(pk.mod.weight) not in arena.offsets:
            continue
        o, n = arena.offsets[id(pk.mod.weight)]
        out.append((o, n, cout, n // (cin * cout), cin, pk))
    out.sort(key=lambda t: t[0])
    return out


class _BNState:
    """Per-slot BatchNorm side tensors."""

    def __init__(self, C, device):
        self.mean = torch.empty(C, dtype=torch.float32, device=device)
        self.rstd = torch.empty(C, dtype=torch.float32, device=device)
        self.scale = torch.empty(C, dtype=torch.float32, device=device)
        self.shift = torch.empty(C, dtype=torch.float32, device=device)
        self.sums = torch.empty(2 * C, dtype=torch.float32, device=device)


def _conv_bn_train(d, x1, x2, wf, bias, z, a, act, bn: nn.BatchNorm2d, st: _BNState, n_updates, stats):
    """Convolution + BatchNorm2d(train) + activation (pai_conv_fwd_bn: one fused finish launch for the bottleneck
    layers, conv -> finalize -> apply for the others)."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    ops.conv_fwd_bn(d, x1, x2, wf, bias, z, a, act, bn.weight, bn.bias, float(bn.eps), float(mom), n_updates,
                    bn.running_mean, bn.running_var, bn.num_batches_tracked, st.mean, st.rstd, st.scale, st.shift, stats)


def _bn_forward(bn: nn.BatchNorm2d, st: _BNState, stats, rows, count, training, n_updates):
    C = bn.num_features
    if training:
        mom = bn.momentum if bn.momentum is not None else 0.1
        ops.bn_finalize(stats, rows, C, count, bn.weight, bn.bias, float(bn.eps), float(mom), n_updates,
                        bn.running_mean, bn.running_var, bn.num_batches_tracked, st.mean, st.rstd,
                        st.scale, st.shift)
    else:
        ops.bn_eval_coeffs(C, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                           st.scale, st.shift)


class _SideStream:
    """Weight gradients are off the critical path of the backward pass (only the input-gradient
    chain is sequential), so they are issued on a second HIP stream and overlap the next layers'
    dgrad / BatchNorm kernels.  ``fork`` orders the side stream after everything issued so far on
    the main stream, ``join`` makes the main stream wait for the side stream."""

    enabled = True   # PAI_NO_OVERLAP=1 turns it off (per-kernel timing without co-scheduling)

    def __init__(self):
        import os
        self.stream = None
        self.stream2 = None
        self.scratch_ev = None
        self._scratch_marked = False
        self.on = _SideStream.enabled and os.environ.get("PAI_NO_OVERLAP", "0") in ("", "0")
        # (round 3 measured a cap on what goes to the side stream, PAI_OVERLAP_GFLOP: 6.33-6.35 ms/step with every weight
        #  gradient there, 6.67-6.89 with only the launches below 100 / 50 / 20 GFLOP -- the large layers carry the gain.
        #  The switch is gone: weight gradients of one pass all run on ONE stream, which is also what keeps the handle's
        #  single weight-gradient slab buffer free of cross-stream races.)
        # (stream priorities measured at batch 64: side stream at high priority 14.2 ms/step, at low priority
        #  7.49, main stream at high priority 7.39 against 7.47 with both at the default -- left at the default)

    def fork(self, d=None):
        if not self.on:
            return torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        ops.stream_wait_last(self.stream, torch.cuda.current_stream())      # C-ABI edge: recorded into launch plans
        return self.stream

    def fork_tail(self):
        """A second side stream for the LAST weight gradient of a backward pass (the thin first layer,
        HBM-bound): it then runs beside the previous layer's MFMA-bound weight gradient instead of behind it,
        which shortens the tail during which the main stream only waits for `join`."""
        if not self.on:
            return torch.cuda.current_stream()
        if self.stream2 is None:
            self.stream2 = torch.cuda.Stream()
        ops.stream_wait_last(self.stream2, torch.cuda.current_stream())
        if self._scratch_marked:
            # the thin weight-gradient kernels share the tail of the registered scratch buffer
            self.scratch_ev.wait(self.stream2)
        return self.stream2

    def mark_scratch(self):
        """Call inside the side-stream context right after a thin-layer weight gradient has been issued."""
        if self.on and self.stream is not None:
            if self.scratch_ev is None:
                self.scratch_ev = ops.Event()
            self.scratch_ev.record(self.stream)
            self._scratch_marked = True

    def join(self):
        if self.on and self.stream is not None:
            ops.stream_wait_last(torch.cuda.current_stream(), self.stream)
        if self.on and self.stream2 is not None:
            ops.stream_wait_last(torch.cuda.current_stream(), self.stream2)


# --------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------
class UnetEngine:
    overwrites_weight_grads = True     # backward(..., fresh=True) writes every conv-weight gradient (see GradArena.begin_backward)

    def __init__(self, unet: nn.Module):
        self.unet = unet
        encs, decs = list(unet.encoders), list(unet.decoders)
        self.L = len(encs)
        assert len(decs) == self.L
        self.enc_conv = [encs[0]] + [e.encode[1] for e in encs[1:]]
        self.enc_bn = [None] + [e.encode[2] if isinstance(e.encode[2], nn.BatchNorm2d) else None
                                for e in encs[1:]]
        self.dec_conv = [d.decode[1] for d in decs[:-1]] + [decs[-1]]
        self.dec_bn = [d.decode[2] for d in decs[:-1]] + [None]
        # nn.Dropout2d of the widest decoder blocks (reference models/pix2pix.py:108,176-179); 0 = Identity
        self.dec_drop = [float(d.decode[3].p) if isinstance(d.decode[3], nn.Dropout2d) else 0.0
                         for d in decs[:-1]] + [0.0]
        # tests inject masks here: fn(j, N, C, p, device) -> fp32 [N, C] of {0, 1 / (1 - p)}
        self.dropout_mask_fn: Optional[Callable] = None
        self.in_ch = self.enc_conv[0].weight.shape[1]
        self.out_ch = self.dec_conv[-1].weight.shape[1]
        self.enc_c = [c.weight.shape[0] for c in self.enc_conv]
        self.dec_c = [c.weight.shape[1] for c in self.dec_conv]
        self.weights_generation = [0]
        self.enc_packs = [_Packs(c, need_dgrad=(i > 0), gen=self.weights_generation)
                          for i, c in enumerate(self.enc_conv)]
        self.dec_packs = [_Packs(c, need_dgrad=True, gen=self.weights_generation) for c in self.dec_conv]
        self._plans = {}
        self._arena: Optional[GradArena] = None
        self._side = _SideStream()
        self.grad_ready_hook: Optional[Callable[[GradArena, int], None]] = None
        self.debug_capture: Optional[dict] = None   # tests: name -> cloned intermediate tensor

    def _dbg(self, name, t):
        if self.debug_capture is not None:
            self.debug_capture[name] = t.detach().clone()

    def _dropout(self, S, j, M, C, dtype):
        """Training-mode Dropout2d behind decoder j's BatchNorm: S['r'][j] *= mask[n][c] in place; the mask
        is kept in the slot for the backward pass.  The Bernoulli draw uses torch's device generator."""
        p = self.dec_drop[j]
        N = S["P"]["N"]
        if self.dropout_mask_fn is not None:
            mask = self.dropout_mask_fn(j, N, C, p, S["r"][j].device).to(torch.float32).contiguous()
        else:
            keep = torch.full((N, C), 1.0 - p, dtype=torch.float32, device=S["r"][j].device)
            mask = torch.bernoulli(keep) / (1.0 - p)
        S["drop"][j] = mask
        ops.dropout2d(dtype, S["r"][j], mask, N, M // N, C, S["r"][j])

    # ---- parameters -------------------------------------------------------------------
    def ordered_params(self):
        """(param, conv module) in backward-completion order."""
        out = []
        L = self.L
        for j in range(L - 1, -1, -1):
            if self.dec_bn[j] is not None:
                out += [(self.dec_bn[j].weight, None), (self.dec_bn[j].bias, None)]
            out += [(self.dec_conv[j].weight, self.dec_conv[j]), (self.dec_conv[j].bias, None)]
        for i in range(L - 1, -1, -1):
            if self.enc_bn[i] is not None:
                out += [(self.enc_bn[i].weight, None), (self.enc_bn[i].bias, None)]
            out += [(self.enc_conv[i].weight, self.enc_conv[i]), (self.enc_conv[i].bias, None)]
        return out

    def arena(self) -> GradArena:
        dev = self.enc_conv[0].weight.device
        if self._arena is None or self._arena.flat.device != dev:
            for c in self.enc_conv + self.dec_conv:
                to_fwd_pack_(c)
            self._arena = GradArena(self.ordered_params(), dev)
        return self._arena

    def pack_targets(self):
        arena = self.arena()
        if getattr(self, "_pt_cache", (None, None))[0] is not arena:
            self._pt_cache = (arena, pack_targets(arena, self.enc_packs + self.dec_packs))
        return self._pt_cache[1]

    def all_packs(self):
        return self.enc_packs + self.dec_packs

    # ---- plan / buffers ------------------------------------------------------------------
    def _plan(self, N, H, W, dtype, device):
        key = (N, H, W, dtype, str(device))
        if key in self._plans:
            return self._plans[key]
        L = self.L
        if H % (1 << L) or W % (1 << L):
            raise ops.PaiError(f"input {H}x{W} must be divisible by 2^{L}")
        P = {"N": N, "H": H, "W": W, "dtype": dtype, "device": device, "slots": [], "free": []}
        eh = [H >> (i + 1) for i in range(L)]
        ew = [W >> (i + 1) for i in range(L)]
        P["eh"], P["ew"] = eh, ew
        P["enc_desc"] = []
        cin = self.in_ch
        for i in range(L):
            hin = H >> i
            win = W >> i
            act = ACT_LRELU if i == 0 else ACT_NONE
            P["enc_desc"].append(ops.make_desc(dtype, 0, N, hin, win, cin, 0, self.enc_c[i], 2, 0, 0, act))
            cin = self.enc_c[i]
        # weight-gradient calls of the last dense layers of the backward pass (encoders[1], [2], ...): the input-gradient
        # chain ends while they run
        P["dec_desc"] = []
        for j in range(L):
            hin, win = eh[L - 1 - j], ew[L - 1 - j]
            if j == 0:
                c1, c2, r1, r2 = self.enc_c[L - 1], 0, 1, 0
            else:
                c1, c2 = self.dec_c[j - 1], self.enc_c[L - 1 - j]
                r1, r2 = 0, 1
            act = ACT_NONE
            if j == L - 1:
                r1, r2, act = 0, 0, ACT_TANH
            P["dec_desc"].append(ops.make_desc(dtype, 1, N, hin, win, c1, c2, self.dec_c[j], 2, r1, r2, act))
        rows = 1   # floats: (partial rows + fp64 reduction scratch) x 2 x C, largest layer
        for i in range(L):
            if self.enc_bn[i] is not None:
                rows = max(rows, ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(P["enc_desc"][i]))
                           * 2 * self.enc_c[i])
        for j in range(L):
            if self.dec_bn[j] is not None:
                rows = max(rows, ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(P["dec_desc"][j]))
                           * 2 * self.dec_c[j])
        P["stats"] = torch.empty(rows, dtype=torch.float32, device=device)
        mx = 1
        for i in range(L):
            if self.enc_bn[i] is not None:
                mx = max(mx, ops.bn_bwd_partial_rows(N * eh[i] * ew[i]) * 2 * self.enc_c[i])
        for j in range(L):
            if self.dec_bn[j] is not None:
                mx = max(mx, ops.bn_bwd_partial_rows(N * 4 * eh[L - 1 - j] * ew[L - 1 - j]) * 2 * self.dec_c[j])
        for i in range(2, L):      # fused first pass of encoder i-1's BatchNorm backward in encoder i's dgrad
            if self.enc_bn[i - 1] is not None:
                mx = max(mx, ops.conv_dgrad_bn_rows_max(P["enc_desc"][i]) * 2 * self.enc_c[i - 1])
        for j in range(1, L):      # ... of decoder j-1's in decoder j's dgrad (j = L-1: the head's thin kernel)
            if self.dec_bn[j - 1] is not None:
                mx = max(mx, ops.conv_dgrad_bn_rows_max(P["dec_desc"][j]) * 2 * self.dec_c[j - 1])
        P["bwd_partials"] = torch.empty(mx, dtype=torch.float32, device=device)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, op) for d in P["enc_desc"] + P["dec_desc"]
                                 for op in (0, 1)), device)
        ops.ensure_scratch(ops.scratch_bytes_for(P["enc_desc"] + P["dec_desc"]), device)
        ops.ensure_wgrad_workspace(P["enc_desc"] + P["dec_desc"], device)
        self._plans[key] = P
        return P

    def _new_slot(self, P):
        L, N, dt, dev = self.L, P["N"], P["dtype"], P["device"]
        eh, ew = P["eh"], P["ew"]

        def buf(h, w, c, dtype=dt):
            return torch.empty(N * h * w * c, dtype=dtype, device=dev)

        S = {"P": P}
        S["x"] = buf(P["H"], P["W"], self.in_ch)
        S["z"] = [buf(eh[i], ew[i], self.enc_c[i]) for i in range(L)]
        S["a"] = [buf(eh[i], ew[i], self.enc_c[i]) if i < L - 1 else None for i in range(L)]
        S["ebn"] = [_BNState(self.enc_c[i], dev) if self.enc_bn[i] is not None else None for i in range(L)]
        dh = [eh[L - 1 - j] * 2 for j in range(L)]
        dw = [ew[L - 1 - j] * 2 for j in range(L)]
        S["dh"], S["dw"] = dh, dw
        S["w"] = [buf(dh[j], dw[j], self.dec_c[j]) if j < L - 1 else None for j in range(L)]
        S["r"] = [buf(dh[j], dw[j], self.dec_c[j]) if j < L - 1 else None for j in range(L)]
        S["dbn"] = [_BNState(self.dec_c[j], dev) if self.dec_bn[j] is not None else None for j in range(L)]
        S["pred"] = torch.empty(N, self.out_ch, P["H"], P["W"], dtype=torch.float32, device=dev)
        S["grads"] = None
        return S

    def _grad_bufs(self, S):
        if S["grads"] is not None:
            return S["grads"]
        P = S["P"]
        L, N, dt, dev = self.L, P["N"], P["dtype"], P["device"]
        eh, ew = P["eh"], P["ew"]
        G = {}
        G["ga"] = [torch.empty_like(S["z"][i]) if i < L - 1 else None for i in range(L)]      # wrt a_i
        G["gskip"] = [torch.empty_like(S["z"][i]) if i < L - 1 else None for i in range(L)]   # skip part
        G["gz_last"] = torch.empty_like(S["z"][L - 1])                                       # wrt relu(z_last)
        G["gr"] = [torch.empty_like(S["w"][j]) if j < L - 1 else None for j in range(L)]      # wrt r_j
        biggest = max([S["z"][i].numel() for i in range(L)] + [S["w"][j].numel() for j in range(L - 1)]
                      + [S["pred"].numel()])
        G["du"] = torch.empty(biggest, dtype=dt, device=dev)
        # dz is read by the side-stream weight gradient while the main stream moves on: one per layer
        G["dz_head"] = torch.empty(S["pred"].numel(), dtype=dt, device=dev)
        G["dz_dec"] = [torch.empty_like(S["w"][j]) if j < L - 1 else None for j in range(L)]
        G["dz_enc"] = [torch.empty_like(S["z"][i]) for i in range(L)]
        S["grads"] = G
        return G

    def acquire(self, N, H, W, dtype, device):
        P = self._plan(N, H, W, dtype, device)
        if P["free"]:
            return P["free"].pop()
        S = self._new_slot(P)
        P["slots"].append(S)
        return S

    def release(self, S):
        S["P"]["free"].append(S)

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, training: bool, bn_updates: int, dtype: torch.dtype):
        """x: fp32 [N, Cin, H, W] on the HIP device.  Returns (pred fp32 [N,Cout,H,W], slot)."""
        if not x.is_cuda:
            raise ops.PaiError("Unet (HIP) needs a HIP device tensor; there is no CPU path")
        N, Ci, H, W = x.shape
        if Ci != self.in_ch:
            raise ops.PaiError(f"expected {self.in_ch} input channels, got {Ci}")
        L = self.L
        S = self.acquire(N, H, W, dtype, x.device)
        P = S["P"]
        S["drop"] = {}
        xs = x.to(torch.float32)
        xs = xs.contiguous() if Ci == 1 else xs.permute(0, 2, 3, 1).contiguous()
        if dtype == torch.float32:
            S["x"] = xs.reshape(-1)
        else:
            ops.cast(xs, S["x"])
        eh, ew = P["eh"], P["ew"]
        refresh_packs(self.enc_packs + self.dec_packs, dtype)

        # encoder 0: bare Conv2d (reference models/pix2pix.py:141-147); raw + LeakyReLU copies
        wf, _ = self.enc_packs[0].get(dtype)
        ops.conv_fwd(P["enc_desc"][0], S["x"], None, wf, self.enc_conv[0].bias, y_raw=S["z"][0],
                     y_act=S["a"][0] if L > 1 else None)
        for i in range(1, L):
            wf, _ = self.enc_packs[i].get(dtype)
            bn = self.enc_bn[i]
            d = P["enc_desc"][i]
            if bn is not None:
                M = N * eh[i] * ew[i]
                if training:
                    _conv_bn_train(d, S["a"][i - 1], None, wf, self.enc_conv[i].bias, S["z"][i], S["a"][i], ACT_LRELU,
                                   bn, S["ebn"][i], bn_updates, P["stats"])
                else:
                    ops.conv_fwd(d, S["a"][i - 1], None, wf, self.enc_conv[i].bias, y_raw=S["z"][i])
                    _bn_forward(bn, S["ebn"][i], None, 0, M, False, 0)
                    ops.bn_apply(dtype, S["z"][i], M, self.enc_c[i], S["ebn"][i].scale, S["ebn"][i].shift,
                                 ACT_LRELU, S["a"][i])
            else:
                ops.conv_fwd(d, S["a"][i - 1], None, wf, self.enc_conv[i].bias, y_raw=S["z"][i])
                if i < L - 1:
                    # norm-less inner encoder (not produced by the reference topology, kept general)
                    raise ops.PaiError("only the last encoder may be norm-free")
        # decoders
        for j in range(L):
            wf, _ = self.dec_packs[j].get(dtype)
            d = P["dec_desc"][j]
            if j == 0:
                x1, x2 = S["z"][L - 1], None
            else:
                x1 = S["r"][j - 1]
                skip = L - 1 - j
                x2 = S["a"][skip] if skip > 0 else S["z"][0]
                if j == L - 1:
                    x2 = S["z"][0]  # raw encoder-0 output, no activation (pix2pix.py:185-193)
            if j < L - 1:
                bn = self.dec_bn[j]
                M = N * S["dh"][j] * S["dw"][j]
                act = ACT_RELU if j < L - 2 else ACT_NONE
                if training:
                    _conv_bn_train(d, x1, x2, wf, self.dec_conv[j].bias, S["w"][j], S["r"][j], act, bn, S["dbn"][j],
                                   bn_updates, P["stats"])
                else:
                    ops.conv_fwd(d, x1, x2, wf, self.dec_conv[j].bias, y_raw=S["w"][j])
                    _bn_forward(bn, S["dbn"][j], None, 0, M, False, 0)
                    ops.bn_apply(dtype, S["w"][j], M, self.dec_c[j], S["dbn"][j].scale, S["dbn"][j].shift, act,
                                 S["r"][j])
                if training and self.dec_drop[j] > 0:
                    # ReLU(Dropout2d(y)) == Dropout2d(ReLU(y)) (mask >= 0): applied to the stored activation
                    self._dropout(S, j, M, self.dec_c[j], dtype)
            else:
                # the prediction is handed to the caller: a fresh tensor per call (the caching allocator makes this a
                # pointer bump), so that a later forward through the same slot cannot overwrite what the caller holds
                S["pred"] = torch.empty_like(S["pred"])
                ops.conv_fwd(d, x1, x2, wf, self.dec_conv[j].bias, y_f32=S["pred"])
        if L == 1:
            raise ops.PaiError("Unet needs at least two levels")
        pred = S["pred"]
        if self.out_ch != 1:
            pred = pred.view(N, H, W, self.out_ch).permute(0, 3, 1, 2)
        return pred, S

    # ---- backward ---------------------------------------------------------------------------
    def backward(self, S, gpred: torch.Tensor, fresh: bool = False):
        """Accumulates every parameter gradient into the arena.  gpred: fp32, pred's shape.  ``fresh``: this is the first
        backward pass since zero_grad and ``GradArena.begin_backward(..., overwrite_weights=True)`` left the conv-weight
        segments un-cleared: every weight (and non-BatchNorm bias) gradient is then WRITTEN (pai_conv_wgrad_overwrite: no
        zero fill, no read of dW) instead of added."""
        P = S["P"]
        L, N, dtype = self.L, P["N"], P["dtype"]
        eh, ew = P["eh"], P["ew"]
        G = self._grad_bufs(S)
        A = self.arena()
        hook = self.grad_ready_hook
        if self.out_ch != 1:
            gpred = gpred.permute(0, 2, 3, 1)
        gpred = gpred.contiguous()
        if gpred.dtype != torch.float32:
            gpred = gpred.float()

        def done(p):
            if hook is not None:
                hook(A, A.end_of(p))

        side = self._side

        # first pass: dense weight gradients are WRITTEN; bias gradients and the thin layers' weights are added to
        # segments GradArena.begin_backward cleared in one launch
        conv_wgrad = ops.conv_wgrad_overwrite_w if fresh else ops.conv_wgrad

        # The weight gradients of the bottleneck layers (<= 1024 output pixels: encoders[5-7], decoders[0-2]) are issued THREE at a
        # time behind one fork: a fork is an event record on the main queue and a wait on the side queue, ~5 us of queue latency
        # each between launches of 8-50 us, and the side stream lags the main one there anyway.  Same box, six interleaved
        # runs each: 5.741 ms/step with a fork per layer, 5.723 in pairs, **5.692** in threes, 5.742 in fours; with the
        # 8 x 8-pixel layers included 5.77-5.81 (PAI_WGRAD_BATCH / PAI_WGRAD_BATCH_PIX: the A/B switches).
        batch_n = int(os.environ.get("PAI_WGRAD_BATCH", "3"))
        held = []

        def flush_held():
            if held:
                with torch.cuda.stream(side.fork()):
                    for run in held:
                        run()
                held.clear()

        def wgrad(d, x1, x2, dz, conv, with_bias):
            """Weight (and bias) gradient of one layer on the side stream, beside its own input gradient.  (Round 4 measured
            the alternatives: holding the first decoders' weight gradients back until the main stream is in the bottleneck
            chain, 6.34-6.48 against 6.25-6.27 ms/step; two workgroups per CU for the last one of a pass, no change.)"""
            def run():
                fn = ops.conv_wgrad if min(_cin_cout(conv)) <= 2 else conv_wgrad
                fn(d, x1, x2, dz, A.seg(conv.weight), A.seg(conv.bias) if with_bias else None)
                done(conv.bias)
            if batch_n > 1 and dz.numel() // max(1, conv.weight.shape[1] if isinstance(conv, nn.ConvTranspose2d) else conv.weight.shape[0]) <= int(os.environ.get("PAI_WGRAD_BATCH_PIX", "1024")):
                held.append(run)
                if len(held) >= batch_n:
                    flush_held()
                return
            flush_held()
            with torch.cuda.stream(side.fork(d)):
                run()

        # head: tanh' then the bare ConvTranspose2d (pix2pix.py:185-193,216)
        j = L - 1
        dh = G["dz_head"]
        ops.tanh_bwd(dtype, S["pred"], gpred, None, dh)
        d = P["dec_desc"][j]
        x1 = S["r"][j - 1]
        x2 = S["z"][0]
        wgrad(d, x1, x2, dh, self.dec_conv[j], True)
        side.mark_scratch()
        _, wd = self.dec_packs[j].get(dtype)
        part = P["bwd_partials"]
        # Every input-gradient launch also runs the first half of the backward of the layer that PRODUCED its input
        # (activation derivative, the encoder/skip sum, BatchNorm-backward partial sums) in its store
        # (pai_conv_dgrad_bn): the gradient tensor it writes is already `du`.  The head's thin kernel does so too
        # (decoders[L-2] is read without an activation: du IS the gradient it stores), unless Dropout2d sits in between.
        fused_rows = None   # None: a plain gradient for decoder L-2 waits in G["gr"][L-2]
        if (j - 1) in S["drop"] or self.dec_bn[j - 1] is None:
            ops.conv_dgrad(d, dh, wd, G["gr"][j - 1], G["gskip"][0])
            if (j - 1) in S["drop"]:
                ops.dropout2d(dtype, G["gr"][j - 1], S["drop"][j - 1], N, S["dh"][j - 1] * S["dw"][j - 1], self.dec_c[j - 1],
                              G["gr"][j - 1])
        else:
            pst, pbn = S["dbn"][j - 1], self.dec_bn[j - 1]
            ops.conv_dgrad_bn_apply(d, dh, wd, G["gr"][j - 1], G["gskip"][0], S["w"][j - 1], ACT_NONE, None, ACT_NONE,
                                    None, None, pst.mean, pst.rstd, part, pbn.weight, pst.sums, A.seg(pbn.weight),
                                    A.seg(pbn.bias), G["dz_dec"][j - 1])
            fused_rows = -1
        # BN decoders
        for j in range(L - 2, -1, -1):
            bn, st, conv = self.dec_bn[j], S["dbn"][j], self.dec_conv[j]
            M = N * S["dh"][j] * S["dw"][j]
            C = self.dec_c[j]
            n = M * C
            act = ACT_RELU if j < L - 2 else ACT_NONE
            dz = G["dz_dec"][j]
            if fused_rows is None:
                # no activation between this BatchNorm and its consumer: du IS the incoming gradient, not a copy of it
                du = G["du"][:n] if act != ACT_NONE else G["gr"][j]
                ops.bn_bwd_reduce(dtype, G["gr"][j], act, None, ACT_NONE, S["r"][j] if act != ACT_NONE else None,
                                  S["w"][j], M, C, st.mean, st.rstd, du if act != ACT_NONE else None, part, st.sums,
                                  A.seg(bn.weight), A.seg(bn.bias))
                ops.bn_bwd_apply(dtype, du, S["w"][j], M, C, st.mean, st.rstd, bn.weight, st.sums, dz)
            # else: the input-gradient call of decoder j+1 carried this layer's BatchNorm backward through to dz
            # (pai_conv_dgrad_bn_apply: sums, dgamma / dbeta and dz are written)
            self._dbg(f"dec{j}.dz", dz)
            d = P["dec_desc"][j]
            if j == 0:
                x1, x2 = S["z"][L - 1], None
            else:
                skip = L - 1 - j
                x1, x2 = S["r"][j - 1], (S["a"][skip] if skip > 0 else S["z"][0])
            # a conv bias in front of a BatchNorm has an identically zero gradient (BN subtracts the
            # batch mean); the arena already holds zeros for it, no reduction pass is spent on it
            wgrad(d, x1, x2, dz, conv, False)
            _, wd = self.dec_packs[j].get(dtype)
            if j == 0:
                # producer: the norm-free last encoder, consumed through ReLU -> dz_last = relu'(z_last) * g
                fused_rows = ops.conv_dgrad_bn(d, dz, wd, G["dz_enc"][L - 1], None, S["z"][L - 1], ACT_RELU)
            elif (j - 1) in S["drop"]:
                # producer carries Dropout2d: plain gradient, mask, then the two-pass BatchNorm backward
                ops.conv_dgrad(d, dz, wd, G["gr"][j - 1], G["gskip"][L - 1 - j])
                ops.dropout2d(dtype, G["gr"][j - 1], S["drop"][j - 1], N, S["dh"][j - 1] * S["dw"][j - 1],
                              self.dec_c[j - 1], G["gr"][j - 1])
                fused_rows = None
            else:
                pst, pbn = S["dbn"][j - 1], self.dec_bn[j - 1]   # producer: decoder j-1 (BatchNorm, read through ReLU)
                ops.conv_dgrad_bn_apply(d, dz, wd, G["gr"][j - 1], G["gskip"][L - 1 - j], S["w"][j - 1], ACT_RELU, None,
                                        ACT_NONE, pst.scale, pst.shift, pst.mean, pst.rstd, part, pbn.weight, pst.sums,
                                        A.seg(pbn.weight), A.seg(pbn.bias), G["dz_dec"][j - 1])
                fused_rows = -1
        # last encoder (no norm)
        i = L - 1
        conv = self.enc_conv[i]
        dz = G["dz_enc"][i]
        d = P["enc_desc"][i]
        wgrad(d, S["a"][i - 1], None, dz, conv, True)
        _, wd = self.enc_packs[i].get(dtype)

        def enc_dgrad(i, dz, wd):
            """Input gradient of encoder i; producer = encoder i-1, read through LeakyReLU here and through
            ReLU (raw for encoder 0) by its skip decoder: du = lrelu'(pre) * g + relu'(pre) * g_skip."""
            d = P["enc_desc"][i]
            if i - 1 == 0:   # bare Conv2d: no norm; the gradient written IS dz of encoder 0
                return ops.conv_dgrad_bn(d, dz, wd, G["dz_enc"][0], None, S["z"][0], ACT_LRELU, G["gskip"][0], ACT_NONE)
            pst, pbn = S["ebn"][i - 1], self.enc_bn[i - 1]
            ops.conv_dgrad_bn_apply(d, dz, wd, G["ga"][i - 1], None, S["z"][i - 1], ACT_LRELU, G["gskip"][i - 1], ACT_RELU,
                                    pst.scale, pst.shift, pst.mean, pst.rstd, part, pbn.weight, pst.sums,
                                    A.seg(pbn.weight), A.seg(pbn.bias), G["dz_enc"][i - 1])
            return -1

        fused_rows = enc_dgrad(i, dz, wd)
        # BN encoders
        for i in range(L - 2, 0, -1):
            bn, st, conv = self.enc_bn[i], S["ebn"][i], self.enc_conv[i]
            M = N * eh[i] * ew[i]
            C = self.enc_c[i]
            dz = G["dz_enc"][i]       # written by encoder i+1's input-gradient call (pai_conv_dgrad_bn_apply)
            d = P["enc_desc"][i]
            # (PAI_HINT_SOLO for the last 1-3 encoders' weight gradients re-measured in round 6, with the round-3 loop and with the
            #  pipelined one: 5.64 / 5.62 ms/step with and without)
            wgrad(P["enc_desc"][i], S["a"][i - 1], None, dz, conv, False)   # bias grad == 0 (BN)
            _, wd = self.enc_packs[i].get(dtype)
            fused_rows = enc_dgrad(i, dz, wd)
        # encoder 0 (its dz came out of encoder 1's input gradient): on the tail stream, beside encoder 1's
        conv0 = self.enc_conv[0]
        flush_held()
        with torch.cuda.stream(side.fork_tail()):
            (ops.conv_wgrad if min(_cin_cout(conv0)) <= 2 else conv_wgrad)(
                P["enc_desc"][0], S["x"], None, G["dz_enc"][0], A.seg(conv0.weight), A.seg(conv0.bias))
        side.join()
        done(conv0.bias)      # both side streams have been joined: the whole arena is final


# --------------------------------------------------------------------------------------
# PatchGAN discriminator
# --------------------------------------------------------------------------------------
class DiscEngine:
    def __init__(self, disc: nn.Module):
        seq = disc.discriminator
        self.convs = [seq[i].block[0] for i in range(4)] + [seq[4]]
        self.in_ch = self.convs[0].weight.shape[1] // 2
        self.chans = [c.weight.shape[0] for c in self.convs]
        self.weights_generation = [0]
        self.packs = [_Packs(c, need_dgrad=True, gen=self.weights_generation) for c in self.convs]
        self._plans = {}
        self._arena = None
        self._side = _SideStream()
        self.grad_ready_hook = None

    def ordered_params(self):
        out = [(self.convs[4].weight, self.convs[4])]
        for k in range(3, -1, -1):
            out += [(self.convs[k].weight, self.convs[k]), (self.convs[k].bias, None)]
        return out

    def arena(self) -> GradArena:
        dev = self.convs[0].weight.device
        if self._arena is None or self._arena.flat.device != dev:
            for c in self.convs:
                to_fwd_pack_(c)
            self._arena = GradArena(self.ordered_params(), dev)
        return self._arena

    def pack_targets(self):
        arena = self.arena()
        if getattr(self, "_pt_cache", (None, None))[0] is not arena:
            self._pt_cache = (arena, pack_targets(arena, self.packs))
        return self._pt_cache[1]

    def all_packs(self):
        return self.packs

    def _plan(self, N, H, W, dtype, device):
        key = (N, H, W, dtype, str(device))
        if key in self._plans:
            return self._plans[key]
        if H % 16 or W % 16 or H < 32 or W < 32:
            raise ops.PaiError(f"discriminator input {H}x{W} must be a multiple of 16 and >= 32")
        P = {"N": N, "H": H, "W": W, "dtype": dtype, "device": device, "free": [], "desc": []}
        c = self.in_ch
        P["desc"].append(ops.make_desc(dtype, 0, N, H, W, c, c, self.chans[0], 2, 0, 0, ACT_LRELU))
        for k in range(1, 4):
            P["desc"].append(ops.make_desc(dtype, 0, N, H >> k, W >> k, self.chans[k - 1], 0, self.chans[k], 2, 0, 0,
                                           ACT_LRELU))
        P["desc"].append(ops.make_desc(dtype, 0, N, H >> 4, W >> 4, self.chans[3], 0, 1, 1, 0, 0, ACT_NONE))
        P["oh"], P["ow"] = (H >> 4) - 1, (W >> 4) - 1
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, op) for d in P["desc"] for op in (0, 1)), device)
        ops.ensure_scratch(ops.scratch_bytes_for(P["desc"]), device)
        ops.ensure_wgrad_workspace(P["desc"], device)
        self._plans[key] = P
        return P

    def acquire(self, N, H, W, dtype, device):
        P = self._plan(N, H, W, dtype, device)
        if P["free"]:
            return P["free"].pop()
        S = {"P": P}
        S["x"] = torch.empty(N * H * W * self.in_ch, dtype=dtype, device=device)
        S["y"] = torch.empty(N * H * W * self.in_ch, dtype=dtype, device=device)
        S["a"] = [torch.empty(N * (H >> (k + 1)) * (W >> (k + 1)) * self.chans[k], dtype=dtype, device=device)
                  for k in range(4)]
        S["logits"] = torch.empty(N, 1, P["oh"], P["ow"], dtype=torch.float32, device=device)
        S["grads"] = None
        return S

    def release(self, S):
        S["P"]["free"].append(S)

    def _to_nhwc(self, t, dst, dtype):
        t = t.to(torch.float32)
        t = t.contiguous() if t.shape[1] == 1 else t.permute(0, 2, 3, 1).contiguous()
        if dtype == torch.float32:
            return t.reshape(-1)
        ops.cast(t, dst)
        return dst

    def forward(self, x, y, dtype, y2=None):
        """y2: a second image batch evaluated against the same conditioning x in the same pass -- the rows of the
        2N-sample batch are [x|y] then [x|y2] (what ``cat([x, x]), cat([y, y2])`` would feed), written straight into
        the engine's input buffers without materialising the concatenations."""
        if not (x.is_cuda and y.is_cuda):
            raise ops.PaiError("Discriminator (HIP) needs HIP device tensors; there is no CPU path")
        N, C, H, W = x.shape
        if C != self.in_ch or y.shape != x.shape or (y2 is not None and y2.shape != x.shape):
            raise ops.PaiError(f"Discriminator expects [N,{self.in_ch},H,W] tensors of one shape, got {tuple(x.shape)} "
                               f"and {tuple(y.shape)}")
        reps = 1 if y2 is None else 2
        S = self.acquire(N * reps, H, W, dtype, x.device)
        P = S["P"]
        if y2 is None and dtype != torch.float32 and C == 1:
            ops.cast_multi([(x.detach().to(torch.float32).contiguous().reshape(-1), S["x"]),
                            (y.detach().to(torch.float32).contiguous().reshape(-1), S["y"])])
            S["xin"], S["yin"] = S["x"], S["y"]
        elif y2 is None:
            S["xin"] = self._to_nhwc(x, S["x"], dtype)
            S["yin"] = self._to_nhwc(y, S["y"], dtype)
        else:
            half = N * H * W * C
            pairs = []
            for src, dst in ((x, S["x"][:half]), (x, S["x"][half:]), (y, S["y"][:half]), (y2, S["y"][half:])):
                t = src.detach().to(torch.float32)
                t = t.contiguous() if C == 1 else t.permute(0, 2, 3, 1).contiguous()
                pairs.append((t.reshape(-1), dst))
            ops.cast_multi(pairs)      # one launch; fp32 too: C-ABI copies (nodes of the launch plan), not torch's copy_
            S["xin"], S["yin"] = S["x"], S["y"]
        refresh_packs(self.packs, dtype)
        wf, _ = self.packs[0].get(dtype)
        ops.conv_fwd(P["desc"][0], S["xin"], S["yin"], wf, self.convs[0].bias, y_act=S["a"][0])
        for k in range(1, 4):
            wf, _ = self.packs[k].get(dtype)
            ops.conv_fwd(P["desc"][k], S["a"][k - 1], None, wf, self.convs[k].bias, y_act=S["a"][k])
        wf, _ = self.packs[4].get(dtype)
        S["logits"] = torch.empty_like(S["logits"])    # owned by the caller, see UnetEngine.forward
        ops.conv_fwd(P["desc"][4], S["a"][3], None, wf, None, y_f32=S["logits"])
        return S["logits"], S

    def _wgrad_desc(self, P, k):
        """The descriptor block k's weight gradient is launched with: PAI_HINT_SOLO -- two workgroups per CU -- in the
        discriminator's own backward pass, whose weight-gradient stream is as long as its input-gradient chain + thin
        first-layer weight gradient (kernel trace: 820 against 625 + 199 us).  Same box, interleaved: 5.680 -> 5.663
        ms/step (PAI_D_WGRAD_SOLO=0 turns it off).  Hints change launch geometry only, never results."""
        if os.environ.get("PAI_D_WGRAD_SOLO", "1") in ("", "0"):
            return P["desc"][k]
        solo = P.setdefault("desc_solo", {})
        if k not in solo:
            import copy
            d = copy.copy(P["desc"][k])
            d.hints = 1
            solo[k] = d
        return solo[k]

    def backward(self, S, glogits, need_params: bool, need_dy: bool, fresh: bool = False):
        """``fresh``: see UnetEngine.backward."""
        conv_wgrad = ops.conv_wgrad_overwrite_w if fresh else ops.conv_wgrad
        P = S["P"]
        N, H, W, dtype, dev = P["N"], P["H"], P["W"], P["dtype"], P["device"]
        if S["grads"] is None:
            G = {"du": [torch.empty_like(a) for a in S["a"]],
                 "dl": torch.empty(S["logits"].numel(), dtype=dtype, device=dev),
                 "dy": torch.empty(N * H * W * self.in_ch, dtype=dtype, device=dev)}
            S["grads"] = G
        G = S["grads"]
        A = self.arena() if need_params else None
        hook = self.grad_ready_hook
        glogits = glogits.contiguous().float()
        if dtype == torch.float32:
            dl = glogits.reshape(-1)
        else:
            ops.cast(glogits, G["dl"])
            dl = G["dl"]
        d = P["desc"][4]
        side = self._side
        if need_params:
            with torch.cuda.stream(side.fork(d)):
                ops.conv_wgrad(d, S["a"][3], None, dl, A.seg(self.convs[4].weight), None)     # thin (one output channel): added
                if hook is not None:
                    hook(A, A.end_of(self.convs[4].weight))
            side.mark_scratch()
        _, wd = self.packs[4].get(dtype)
        # du[k] = LeakyReLU'(a[k]) * dgrad of block k+1: the activation backward rides on the dgrad store
        ops.conv_dgrad_act(d, dl, wd, G["du"][3], None, S["a"][3], ACT_LRELU)
        for k in range(3, -1, -1):
            conv = self.convs[k]
            d = P["desc"][k]
            if need_params and k == 0:
                # thin first layer: tail stream, beside block 1's weight gradient; its hook fires after the join
                with torch.cuda.stream(side.fork_tail()):
                    (ops.conv_wgrad if min(_cin_cout(conv)) <= 2 else conv_wgrad)(
                        d, S["xin"], S["yin"], G["du"][0], A.seg(conv.weight), A.seg(conv.bias))
            elif need_params:
                with torch.cuda.stream(side.fork(d)):
                    conv_wgrad(self._wgrad_desc(P, k), S["a"][k - 1], None, G["du"][k], A.seg(conv.weight), A.seg(conv.bias))
                    if hook is not None:
                        hook(A, A.end_of(conv.bias))
            if k > 0:
                _, wd = self.packs[k].get(dtype)
                ops.conv_dgrad_act(d, G["du"][k], wd, G["du"][k - 1], None, S["a"][k - 1], ACT_LRELU)
            elif need_dy:
                _, wd = self.packs[0].get(dtype)
                ops.conv_dgrad(d, G["du"][0], wd, None, G["dy"], only_c2=True)
        side.join()
        if need_params and hook is not None:
            hook(A, A.end_of(self.convs[0].bias))
        if not need_dy:
            return None
        gy = torch.empty(N * H * W * self.in_ch, dtype=torch.float32, device=dev)
        ops.cast(G["dy"], gy)
        if self.in_ch == 1:
            return gy.view(N, 1, H, W)
        return gy.view(N, H, W, self.in_ch).permute(0, 3, 1, 2)
