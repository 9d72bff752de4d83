"""Attention U-Net executor: ``UnetEngine`` with an attention gate on every skip connection
(reference models/attention_unet.py:64-96, wired at :168-170 and :197-206).

Differences from the Pix2Pix schedule (engine.UnetEngine):
  * the gates read the encoder outputs and the decoder outputs UN-activated, so the BatchNorm output
    of every skip-carrying encoder is stored both raw (``b``) and behind LeakyReLU (``a``), and the
    decoder outputs are stored un-activated (their consumers apply ReLU while loading);
  * decoder j >= 1 reads ``cat([r_{j-1}, s_j])`` with ``s_j = x * att`` from gate j-1;
  * in the backward pass the gradient of a gated skip flows through ``pai_gate_*`` and the two
    pointwise convolutions of the gate; its two ends are folded into the producer-backward store of
    those convolutions' input-gradient launches (``pai_conv_dgrad_bn``), so no separate add /
    activation-backward / BatchNorm-reduce pass is spent on them.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .engine import GradArena, UnetEngine, _BNState, _Packs, _bn_forward, to_fwd_pack_
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH


class _Gate:
    """Parameter handles of one AttentionBlock (reference models/attention_unet.py:64-86)."""

    def __init__(self, block: nn.Module, gen):
        self.conv_i, self.bn_i = block.input_gate[0], block.input_gate[1]
        self.conv_s, self.bn_s = block.signal_gate[0], block.signal_gate[1]
        self.conv_a, self.bn_a = block.attention[0], block.attention[1]
        self.C = self.conv_i.weight.shape[1]
        self.K = self.conv_i.weight.shape[0]
        self.pack_i = _Packs(self.conv_i, need_dgrad=True, gen=gen)
        self.pack_s = _Packs(self.conv_s, need_dgrad=True, gen=gen)

    def ordered_params(self):
        """Backward-completion order: the K -> 1 head first, the two C -> K convolutions last."""
        return [(self.conv_a.weight, None), (self.conv_a.bias, None), (self.bn_a.weight, None), (self.bn_a.bias, None),
                (self.bn_i.weight, None), (self.bn_i.bias, None), (self.bn_s.weight, None), (self.bn_s.bias, None),
                (self.conv_i.weight, None), (self.conv_i.bias, None), (self.conv_s.weight, None),
                (self.conv_s.bias, None)]


class AttentionUnetEngine(UnetEngine):
    overwrites_weight_grads = True     # dense U-Net weight gradients are written; the gates' small segments are cleared and added to

    def __init__(self, unet: nn.Module):
        super().__init__(unet)
        # attention_blocks[k] gates the skip of decoder k+1 (reference :199-203)
        self.gates = [_Gate(b, self.weights_generation) for b in unet.attention_blocks]
        assert len(self.gates) == self.L - 1

    def all_packs(self):
        return self.enc_packs + self.dec_packs + [pk for g in self.gates for pk in (g.pack_i, g.pack_s)]

    # gate of decoder j (j >= 1) and the encoder level whose output it gates
    def _gate(self, j):
        return self.gates[j - 1]

    def ordered_params(self):
        out = []
        L = self.L
        for j in range(L - 1, -1, -1):
            if self.dec_bn[j] is not None:
                out += [(self.dec_bn[j].weight, None), (self.dec_bn[j].bias, None)]
            out += [(self.dec_conv[j].weight, self.dec_conv[j]), (self.dec_conv[j].bias, None)]
            if j >= 1:
                out += self._gate(j).ordered_params()
        for i in range(L - 1, -1, -1):
            if self.enc_bn[i] is not None:
                out += [(self.enc_bn[i].weight, None), (self.enc_bn[i].bias, None)]
            out += [(self.enc_conv[i].weight, self.enc_conv[i]), (self.enc_conv[i].bias, None)]
        return out

    # ---- plan / buffers ---------------------------------------------------------------------
    def _plan(self, N, H, W, dtype, device):
        key = (N, H, W, dtype, str(device))
        if key in self._plans:
            return self._plans[key]
        P = super()._plan(N, H, W, dtype, device)
        L = self.L
        eh, ew = P["eh"], P["ew"]
        # decoders read their first input un-activated with ReLU on load; the head reads both raw
        P["dec_desc"] = []
        for j in range(L):
            hin, win = eh[L - 1 - j], ew[L - 1 - j]
            if j == 0:
                c1, c2, r1, r2 = self.enc_c[L - 1], 0, 1, 0
            else:
                c1, c2, r1, r2 = self.dec_c[j - 1], self.enc_c[L - 1 - j], 1, 1
            act = ACT_NONE
            if j == L - 1:
                r1, r2, act = 0, 0, ACT_TANH
            P["dec_desc"].append(ops.make_desc(dtype, 1, N, hin, win, c1, c2, self.dec_c[j], 2, r1, r2, act))
        P["gate_desc"] = [None]
        rows = P["stats"].numel()
        mx = P["bwd_partials"].numel()
        for j in range(1, L):
            g = self._gate(j)
            lvl = L - 1 - j
            d = ops.make_desc(dtype, 0, N, eh[lvl], ew[lvl], g.C, 0, g.K, 1, 0, 0, ACT_NONE, kernel=1)
            P["gate_desc"].append(d)
            M = N * eh[lvl] * ew[lvl]
            rows = max(rows, ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * g.K,
                       ops.bn_stats_buffer_rows(ops.gate_partial_rows(M)) * 2)
            mx = max(mx, ops.gate_partial_rows(M) * 2 * g.K, ops.conv_dgrad_bn_rows_max(d) * 2 * g.C)
        P["stats"] = torch.empty(rows, dtype=torch.float32, device=device)
        P["bwd_partials"] = torch.empty(mx, dtype=torch.float32, device=device)
        P["bwd_partials2"] = torch.empty(mx, dtype=torch.float32, device=device)
        ops.ensure_workspace(max(ops.conv_workspace_bytes(d, op) for d in P["dec_desc"] + P["gate_desc"][1:]
                                 for op in (0, 1)), device)
        ops.ensure_scratch(ops.scratch_bytes_for(P["enc_desc"] + P["dec_desc"]), device)
        ops.ensure_wgrad_workspace(P["enc_desc"] + P["dec_desc"], device)
        return P

    def _new_slot(self, P):
        S = super()._new_slot(P)
        L, N, dt, dev = self.L, P["N"], P["dtype"], P["device"]
        eh, ew = P["eh"], P["ew"]
        # raw BatchNorm output of the skip-carrying encoders (level 0 is the bare conv: its z is the skip)
        S["b"] = [None] + [torch.empty_like(S["z"][i]) if i < L - 1 else None for i in range(1, L)]
        S["gate"] = [None]
        for j in range(1, L):
            g = self._gate(j)
            lvl = L - 1 - j
            M = N * eh[lvl] * ew[lvl]
            f32 = dict(dtype=torch.float32, device=dev)
            S["gate"].append({
                "M": M,
                "ig": torch.empty(M * g.K, dtype=dt, device=dev), "sg": torch.empty(M * g.K, dtype=dt, device=dev),
                "h": torch.empty(M * g.K, dtype=dt, device=dev), "s": torch.empty(M * g.C, dtype=dt, device=dev),
                "logit": torch.empty(M, **f32), "att": torch.empty(M, **f32),
                "bn_i": _BNState(g.K, dev), "bn_s": _BNState(g.K, dev), "bn_a": _BNState(1, dev),
            })
        return S

    def _grad_bufs(self, S):
        if S["grads"] is not None:
            return S["grads"]
        G = super()._grad_bufs(S)
        P = S["P"]
        L, dt, dev = self.L, P["dtype"], P["device"]
        G["gs"] = [None] + [torch.empty_like(S["gate"][j]["s"]) for j in range(1, L)]       # wrt (relu of) s_j
        G["gx"] = [None] + [torch.empty_like(S["gate"][j]["s"]) for j in range(1, L)]       # gate -> wrt x
        G["dxs"] = [None] + [torch.empty_like(S["gate"][j]["s"]) for j in range(1, L)]
        G["dl"] = [None] + [torch.empty_like(S["gate"][j]["logit"]) for j in range(1, L)]
        G["dsum"] = [None] + [torch.empty_like(S["gate"][j]["h"]) for j in range(1, L)]
        G["dig"] = [None] + [torch.empty_like(S["gate"][j]["h"]) for j in range(1, L)]
        G["dsg"] = [None] + [torch.empty_like(S["gate"][j]["h"]) for j in range(1, L)]
        G["gr_raw"] = [torch.empty_like(t) if t is not None else None for t in G["gr"]]   # decoder j's part of d r_{j-1}
        return G

    def _skip(self, S, lvl):
        """Un-activated encoder output of level `lvl` (what ``feats`` holds in the reference, :190-191)."""
        return S["z"][0] if lvl == 0 else S["b"][lvl]

    # ---- forward --------------------------------------------------------------------------------
    def _gate_forward(self, S, j, x, signal, training, bn_updates, dtype):
        g, gs, P = self._gate(j), S["gate"][j], S["P"]
        d, M = P["gate_desc"][j], gs["M"]
        for conv, bn, pack, src, dst, st in ((g.conv_i, g.bn_i, g.pack_i, x, gs["ig"], gs["bn_i"]),
                                             (g.conv_s, g.bn_s, g.pack_s, signal, gs["sg"], gs["bn_s"])):
            wf, _ = pack.get(dtype)
            if training:
                rows = ops.conv_fwd_stats_rows(d)
                ops.conv_fwd(d, src, None, wf, conv.bias, y_raw=dst, stats=P["stats"])
                _bn_forward(bn, st, P["stats"], rows, M, True, bn_updates)
            else:
                ops.conv_fwd(d, src, None, wf, conv.bias, y_raw=dst)
                _bn_forward(bn, st, None, 0, M, False, 0)
        bi, bs, ba = gs["bn_i"], gs["bn_s"], gs["bn_a"]
        ops.gate_hidden(dtype, gs["ig"], gs["sg"], M, g.K, bi.scale, bi.shift, bs.scale, bs.shift, g.conv_a.weight,
                        g.conv_a.bias, gs["h"], gs["logit"], P["stats"])
        _bn_forward(g.bn_a, ba, P["stats"] if training else None, ops.gate_partial_rows(M) if training else 0, M,
                    training, bn_updates)
        ops.gate_apply(dtype, x, gs["logit"], M, g.C, ba.scale, ba.shift, gs["s"], gs["att"])
        return gs["s"]

    def forward(self, x: torch.Tensor, training: bool, bn_updates: int, dtype: torch.dtype):
        if not x.is_cuda:
            raise ops.PaiError("AttentionUnet (HIP) needs a HIP device tensor; there is no CPU path")
        N, Ci, H, W = x.shape
        if Ci != self.in_ch:
            raise ops.PaiError(f"expected {self.in_ch} input channels, got {Ci}")
        L = self.L
        if L < 2:
            raise ops.PaiError("Unet needs at least two levels")
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
        wf, _ = self.enc_packs[0].get(dtype)
        ops.conv_fwd(P["enc_desc"][0], S["x"], None, wf, self.enc_conv[0].bias, y_raw=S["z"][0], y_act=S["a"][0])
        for i in range(1, L):
            wf, _ = self.enc_packs[i].get(dtype)
            bn, d = self.enc_bn[i], P["enc_desc"][i]
            if bn is None:
                if i < L - 1:
                    raise ops.PaiError("only the last encoder may be norm-free")
                ops.conv_fwd(d, S["a"][i - 1], None, wf, self.enc_conv[i].bias, y_raw=S["z"][i])
                continue
            M = N * eh[i] * ew[i]
            if training:
                rows = ops.conv_fwd_stats_rows(d)
                ops.conv_fwd(d, S["a"][i - 1], None, wf, self.enc_conv[i].bias, y_raw=S["z"][i], stats=P["stats"])
                _bn_forward(bn, S["ebn"][i], P["stats"], rows, M, True, bn_updates)
            else:
                ops.conv_fwd(d, S["a"][i - 1], None, wf, self.enc_conv[i].bias, y_raw=S["z"][i])
                _bn_forward(bn, S["ebn"][i], None, 0, M, False, 0)
            st = S["ebn"][i]
            ops.bn_apply(dtype, S["z"][i], M, self.enc_c[i], st.scale, st.shift, ACT_LRELU, S["a"][i])
            ops.bn_apply(dtype, S["z"][i], M, self.enc_c[i], st.scale, st.shift, ACT_NONE, S["b"][i])
        for j in range(L):
            wf, _ = self.dec_packs[j].get(dtype)
            d = P["dec_desc"][j]
            if j == 0:
                x1, x2 = S["z"][L - 1], None
            else:
                x1 = S["r"][j - 1]
                x2 = self._gate_forward(S, j, self._skip(S, L - 1 - j), x1, training, bn_updates, dtype)
            if j < L - 1:
                bn = self.dec_bn[j]
                M = N * S["dh"][j] * S["dw"][j]
                if training:
                    rows = ops.conv_fwd_stats_rows(d)
                    ops.conv_fwd(d, x1, x2, wf, self.dec_conv[j].bias, y_raw=S["w"][j], stats=P["stats"])
                    _bn_forward(bn, S["dbn"][j], P["stats"], rows, M, True, bn_updates)
                else:
                    ops.conv_fwd(d, x1, x2, wf, self.dec_conv[j].bias, y_raw=S["w"][j])
                    _bn_forward(bn, S["dbn"][j], None, 0, M, False, 0)
                # stored un-activated: the next decoder applies ReLU on load, the gate reads it raw
                ops.bn_apply(dtype, S["w"][j], M, self.dec_c[j], S["dbn"][j].scale, S["dbn"][j].shift, ACT_NONE,
                             S["r"][j])
                if training and self.dec_drop[j] > 0:
                    self._dropout(S, j, M, self.dec_c[j], dtype)
            else:
                # the prediction is handed to the caller: a fresh tensor per call (the caching allocator makes this a
                # pointer bump), so that a later forward through the same slot cannot overwrite what the caller holds
                S["pred"] = torch.empty_like(S["pred"])
                ops.conv_fwd(d, x1, x2, wf, self.dec_conv[j].bias, y_f32=S["pred"])
        pred = S["pred"]
        if self.out_ch != 1:
            pred = pred.view(N, H, W, self.out_ch).permute(0, 3, 1, 2)
        return pred, S

    # ---- backward -------------------------------------------------------------------------------
    def backward(self, S, gpred: torch.Tensor, fresh: bool = False):
        """``fresh``: see UnetEngine.backward -- the 4 x 4 layers' weight gradients are then WRITTEN
        (pai_conv_wgrad_overwrite_w); the gates' parameters (registered without a conv module: GradArena._small) and the
        thin head / first layer keep adding into segments begin_backward cleared."""
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
        side = self._side
        part, part2 = P["bwd_partials"], P["bwd_partials2"]

        def done(p):
            if hook is not None:
                hook(A, A.end_of(p))

        def wgrad(d, x1, x2, dz, conv, with_bias, last=None):
            cin, cout = conv.weight.shape[1], conv.weight.shape[0]
            dense = fresh and conv.weight.shape[2] == 4 and min(cin, cout) > 2       # a 4 x 4 U-Net layer, not a gate's 1 x 1
            with torch.cuda.stream(side.fork(d)):
                (ops.conv_wgrad_overwrite_w if dense else ops.conv_wgrad)(
                    d, x1, x2, dz, A.seg(conv.weight), A.seg(conv.bias) if with_bias else None)
                done(last if last is not None else conv.bias)

        def gate_backward(j, relu_out):
            """Gradient of gate j from G['gs'][j] (w.r.t. s_j, before the consumer's ReLU mask).  Leaves the
            gate's contribution to x in G['gx'][j]; returns the number of BatchNorm partial rows its signal
            input-gradient launch wrote for decoder j-1 (whose `du` is then in G['gr'][j-1])."""
            g, gs = self._gate(j), S["gate"][j]
            d, M = P["gate_desc"][j], gs["M"]
            lvl = L - 1 - j
            x, signal = self._skip(S, lvl), S["r"][j - 1]
            ba, bi, bs = gs["bn_a"], gs["bn_i"], gs["bn_s"]
            ops.gate_apply_bwd(dtype, G["gs"][j], x, gs["att"], gs["logit"], M, g.C, ba.mean, ba.rstd, G["dxs"][j],
                               G["dl"][j], part, relu_out)
            ops.bn_bwd_finalize(part, ops.gate_partial_rows(M), 1, ba.sums, A.seg(g.bn_a.weight), A.seg(g.bn_a.bias))
            ops.gate_hidden_bwd(dtype, G["dl"][j], gs["logit"], gs["h"], gs["ig"], gs["sg"], M, g.K, ba.mean, ba.rstd,
                                g.bn_a.weight, ba.sums, g.conv_a.weight, bi.mean, bi.rstd, bs.mean, bs.rstd,
                                G["dsum"][j], part, part2, A.seg(g.conv_a.weight), A.seg(g.conv_a.bias))
            rows = ops.gate_partial_rows(M)
            ops.bn_bwd_finalize(part, rows, g.K, bi.sums, A.seg(g.bn_i.weight), A.seg(g.bn_i.bias))
            ops.bn_bwd_finalize(part2, rows, g.K, bs.sums, A.seg(g.bn_s.weight), A.seg(g.bn_s.bias))
            ops.bn_bwd_apply(dtype, G["dsum"][j], gs["ig"], M, g.K, bi.mean, bi.rstd, g.bn_i.weight, bi.sums, G["dig"][j])
            ops.bn_bwd_apply(dtype, G["dsum"][j], gs["sg"], M, g.K, bs.mean, bs.rstd, g.bn_s.weight, bs.sums, G["dsg"][j])
            # pointwise-conv biases sit in front of a BatchNorm: zero gradient, nothing launched for them
            wgrad(d, x, None, G["dig"][j], g.conv_i, False)
            wgrad(d, signal, None, G["dsg"][j], g.conv_s, False)
            _, wd_i = g.pack_i.get(dtype)
            _, wd_s = g.pack_s.get(dtype)
            # d x (gate part) = W_i^T dig + dout * att
            ops.conv_dgrad_bn(d, G["dig"][j], wd_i, G["gx"][j], None, x, ACT_NONE, G["dxs"][j], ACT_NONE)
            # d r_{j-1} = W_s^T dsg + act'(r) * (gradient from decoder j), plus decoder j-1's BatchNorm sums
            pst = S["dbn"][j - 1]
            if (j - 1) in S["drop"]:
                # decoder j-1 carries Dropout2d: sum the two parts (sign from the stored r), mask, and leave the
                # BatchNorm backward to the two-pass form
                ops.conv_dgrad_bn(d, G["dsg"][j], wd_s, G["gr"][j - 1], None, S["r"][j - 1], ACT_NONE,
                                  G["gr_raw"][j - 1], ACT_RELU if relu_out else ACT_NONE)
                ops.dropout2d(dtype, G["gr"][j - 1], S["drop"][j - 1], N, S["dh"][j - 1] * S["dw"][j - 1],
                              self.dec_c[j - 1], G["gr"][j - 1])
                return None
            return ops.conv_dgrad_bn(d, G["dsg"][j], wd_s, G["gr"][j - 1], None, S["w"][j - 1], ACT_NONE,
                                     G["gr_raw"][j - 1], ACT_RELU if relu_out else ACT_NONE, pst.scale, pst.shift,
                                     pst.mean, pst.rstd, part)

        # head (reference :149-158 of pix2pix.py layout, :213-221 here): tanh', bare ConvTranspose2d, raw inputs
        j = L - 1
        dh = G["dz_head"]
        ops.tanh_bwd(dtype, S["pred"], gpred, None, dh)
        d = P["dec_desc"][j]
        wgrad(d, S["r"][j - 1], S["gate"][j]["s"], dh, self.dec_conv[j], True)
        side.mark_scratch()
        _, wd = self.dec_packs[j].get(dtype)
        ops.conv_dgrad(d, dh, wd, G["gr_raw"][j - 1], G["gs"][j])
        fused_rows = gate_backward(j, relu_out=False)
        for j in range(L - 2, -1, -1):
            bn, st, conv = self.dec_bn[j], S["dbn"][j], self.dec_conv[j]
            M = N * S["dh"][j] * S["dw"][j]
            C = self.dec_c[j]
            dz = G["dz_dec"][j]
            du = G["gr"][j]
            if fused_rows is None:      # Dropout2d layer: G['gr'][j] is the masked gradient w.r.t. BN's output
                du = G["du"][:M * C]
                ops.bn_bwd_reduce(dtype, G["gr"][j], ACT_NONE, None, ACT_NONE, None, S["w"][j], M, C, st.mean, st.rstd,
                                  du, part, st.sums, A.seg(bn.weight), A.seg(bn.bias))
            else:
                ops.bn_bwd_finalize(part, fused_rows, C, st.sums, A.seg(bn.weight), A.seg(bn.bias))
            ops.bn_bwd_apply(dtype, du, S["w"][j], M, C, st.mean, st.rstd, bn.weight, st.sums, dz)
            d = P["dec_desc"][j]
            _, wd = self.dec_packs[j].get(dtype)
            if j == 0:
                wgrad(d, S["z"][L - 1], None, dz, conv, False)
                fused_rows = ops.conv_dgrad_bn(d, dz, wd, G["dz_enc"][L - 1], None, S["z"][L - 1], ACT_RELU)
            else:
                wgrad(d, S["r"][j - 1], S["gate"][j]["s"], dz, conv, False)
                ops.conv_dgrad(d, dz, wd, G["gr_raw"][j - 1], G["gs"][j])
                fused_rows = gate_backward(j, relu_out=True)
        # encoders: as the Pix2Pix schedule, the skip gradient being the gate's (no activation on that path)
        i = L - 1
        dz = G["dz_enc"][i]
        wgrad(P["enc_desc"][i], S["a"][i - 1], None, dz, self.enc_conv[i], True)
        _, wd = self.enc_packs[i].get(dtype)

        def enc_dgrad(i, dz, wd):
            d = P["enc_desc"][i]
            gx = G["gx"][L - 1 - (i - 1)]      # gate of decoder j = L-1-level
            if i - 1 == 0:
                return ops.conv_dgrad_bn(d, dz, wd, G["dz_enc"][0], None, S["z"][0], ACT_LRELU, gx, ACT_NONE)
            pst = S["ebn"][i - 1]
            return ops.conv_dgrad_bn(d, dz, wd, G["ga"][i - 1], None, S["z"][i - 1], ACT_LRELU, gx, ACT_NONE,
                                     pst.scale, pst.shift, pst.mean, pst.rstd, part)

        fused_rows = enc_dgrad(i, dz, wd)
        for i in range(L - 2, 0, -1):
            bn, st, conv = self.enc_bn[i], S["ebn"][i], self.enc_conv[i]
            M = N * eh[i] * ew[i]
            C = self.enc_c[i]
            du, dz = G["ga"][i], G["dz_enc"][i]
            ops.bn_bwd_finalize(part, fused_rows, C, st.sums, A.seg(bn.weight), A.seg(bn.bias))
            ops.bn_bwd_apply(dtype, du, S["z"][i], M, C, st.mean, st.rstd, bn.weight, st.sums, dz)
            wgrad(P["enc_desc"][i], S["a"][i - 1], None, dz, conv, False)
            _, wd = self.enc_packs[i].get(dtype)
            fused_rows = enc_dgrad(i, dz, wd)
        conv0 = self.enc_conv[0]
        with torch.cuda.stream(side.fork_tail()):
            # same selection as the local wgrad(): with in_channels >= 3 encoder 0 is not a thin layer, its segment is NOT
            # among the cleared small ones (GradArena._small) and must be overwritten on the first pass, not added to
            c0_in, c0_out = conv0.weight.shape[1], conv0.weight.shape[0]
            (ops.conv_wgrad_overwrite_w if (fresh and min(c0_in, c0_out) > 2) else ops.conv_wgrad)(
                P["enc_desc"][0], S["x"], None, G["dz_enc"][0], A.seg(conv0.weight), A.seg(conv0.bias))
        side.join()
        done(conv0.bias)
