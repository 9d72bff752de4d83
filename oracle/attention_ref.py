"""Oracle: functional CPU restatement of the reference Attention U-Net generator.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates ``AttentionUnet`` /
``AttentionBlock`` (models/attention_unet.py:48-221; paths relative to /root/reference).  State
keys and shapes are those of ``AttentionUnet.state_dict()``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import torch
import torch.nn.functional as F

from .pix2pix_ref import _bn, _bn_entries, apply_dropout2d, dropout_rates, make_unet_state


def make_attention_unet_state(in_channels: int = 1, out_channels: int = 1,
                              channel_mults: Sequence[int] = (1, 2, 4, 8, 8, 8, 8, 8)) -> OrderedDict:
    """Keys/shapes of ``AttentionUnet.state_dict()``: the Pix2Pix ``Unet`` entries (same encoder /
    decoder construction, attention_unet.py:127-192) followed by ``attention_blocks.k`` for
    k = 0..L-2, block k gating the skip of decoder k+1 with channels = that decoder's input
    signal width (attention_unet.py:176-178: AttentionBlock(channels, channels, channels // 2))."""
    st = make_unet_state(in_channels, out_channels, channel_mults)
    L = len(channel_mults)
    k = 0
    for level in reversed(range(L - 1)):
        c = channel_mults[level] * 64
        a = c // 2
        p = f"attention_blocks.{k}"
        st[p + ".input_gate.0.weight"] = torch.zeros(a, c, 1, 1)
        st[p + ".input_gate.0.bias"] = torch.zeros(a)
        _bn_entries(p + ".input_gate.1", a, st)
        st[p + ".signal_gate.0.weight"] = torch.zeros(a, c, 1, 1)
        st[p + ".signal_gate.0.bias"] = torch.zeros(a)
        _bn_entries(p + ".signal_gate.1", a, st)
        st[p + ".attention.0.weight"] = torch.zeros(1, a, 1, 1)
        st[p + ".attention.0.bias"] = torch.zeros(1)
        _bn_entries(p + ".attention.1", 1, st)
        k += 1
    return st


def attention_block(st, p: str, x, signal, training: bool):
    """``AttentionBlock.forward`` (attention_unet.py:88-96)."""
    h_input = _bn(st, p + ".input_gate.1",
                  F.conv2d(x, st[p + ".input_gate.0.weight"], st[p + ".input_gate.0.bias"]), training)
    h_signal = _bn(st, p + ".signal_gate.1",
                   F.conv2d(signal, st[p + ".signal_gate.0.weight"], st[p + ".signal_gate.0.bias"]), training)
    h = F.relu(h_signal + h_input)
    a = _bn(st, p + ".attention.1",
            F.conv2d(h, st[p + ".attention.0.weight"], st[p + ".attention.0.bias"]), training)
    return x * torch.sigmoid(a)


def attention_unet_forward(st: OrderedDict, x: torch.Tensor, training: bool = True, return_feats: bool = False,
                           dropout: float = 0.0, mask_log=None, masks=None):
    """``AttentionUnet.forward`` (attention_unet.py:194-221): the Pix2Pix encoder (skips are the
    un-activated block outputs), then for every decoder but the first
    ``h = cat([h, attention_blocks[index-1](feats.pop(), h)])`` (:200-203), tanh at the end."""
    L = 1 + sum(1 for k in st if k.startswith("encoders.") and k.endswith("encode.1.weight"))
    h = x if x.dtype == torch.float64 else x.to(torch.float32)   # fp64 only for noise-floor studies
    feats = []
    acts = {}
    h = F.conv2d(h, st["encoders.0.weight"], st["encoders.0.bias"], stride=2, padding=1)
    feats.append(h)
    acts["enc0"] = h
    for i in range(1, L):
        p = f"encoders.{i}.encode"
        h = F.leaky_relu(h, 0.2)
        h = F.conv2d(h, st[p + ".1.weight"], st[p + ".1.bias"], stride=2, padding=1)
        if (p + ".2.weight") in st:
            h = _bn(st, p + ".2", h, training)
        feats.append(h)
        acts[f"enc{i}"] = h
    feats.pop()
    drops = dropout_rates(st, dropout)
    for j in range(L):
        if j != 0:
            s = attention_block(st, f"attention_blocks.{j - 1}", feats.pop(), h, training)
            acts[f"gate{j}"] = s
            h = torch.cat([h, s], dim=1)
        if j < L - 1:
            p = f"decoders.{j}.decode"
            h = F.relu(h)
            h = F.conv_transpose2d(h, st[p + ".1.weight"], st[p + ".1.bias"], stride=2, padding=1)
            h = _bn(st, p + ".2", h, training)
            if training and drops[j] > 0:
                h = apply_dropout2d(h, drops[j], j, mask_log, masks)
            acts[f"dec{j}"] = h
        else:
            h = F.conv_transpose2d(h, st[f"decoders.{j}.weight"], st[f"decoders.{j}.bias"], stride=2, padding=1)
    out = torch.tanh(h)
    if return_feats:
        return out, acts
    return out
