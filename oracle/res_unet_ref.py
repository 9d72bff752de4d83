"""Oracle: functional CPU restatement of the reference residual U-Net family.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates ``ResUnet`` and its blocks (models/res_unet.py:52-334;
paths relative to /root/reference) for res_type "18", "50", "next" and "v2".  State keys / shapes are those of
``ResUnet.state_dict()``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import torch
import torch.nn.functional as F

from .pix2pix_ref import _bn, _bn_entries, apply_dropout2d

# conv_block layouts as index -> kind ('c' conv, 'b' BatchNorm2d, 'r' ReLU); post = ReLU behind the residual sum
BLOCKS = {
    "18": ("cbrcb", True),          # models/res_unet.py:58-64,71
    "50": ("cbrcbrcb", True),       # :86-95,102
    "next": ("cbrcbrcbr", False),   # :147-163 (last ReLU inside conv_block, no post-sum ReLU: SURVEY Q18)
    "v2": ("brcbrc", False),        # :114-121 pre-activation; the skip is BatchNorm -> ReLU -> 1x1 conv (:123-127)
}


def _block_state(st, p, res_type, cin, cout):
    if res_type == "18":
        convs = [(cin, cout, 3, 1), (cout, cout, 3, 1)]
    elif res_type == "50":
        b = cin // 4
        convs = [(cin, b, 1, 1), (b, b, 3, 1), (b, cout, 1, 1)]
    elif res_type == "next":
        w = 4 * 32                                               # bottleneck * cardinality (:144)
        convs = [(cin, w, 1, 1), (w, w, 3, 32), (w, cout, 1, 1)]
    elif res_type == "v2":
        convs = [(cin, cout, 3, 1), (cout, cout, 3, 1)]
    else:
        raise ValueError(res_type)
    layout, _ = BLOCKS[res_type]
    ci = 0
    last = cin
    for idx, kind in enumerate(layout):
        if kind == "c":
            a, b_, k, g = convs[ci]
            st[f"{p}.conv_block.{idx}.weight"] = torch.zeros(b_, a // g, k, k)
            st[f"{p}.conv_block.{idx}.bias"] = torch.zeros(b_)
            last = b_
            ci += 1
        elif kind == "b":
            _bn_entries(f"{p}.conv_block.{idx}", last, st)
    if cin != cout and res_type == "v2":                         # :123-127
        _bn_entries(f"{p}.conv_skip.0", cin, st)
        st[f"{p}.conv_skip.2.weight"] = torch.zeros(cout, cin, 1, 1)
        st[f"{p}.conv_skip.2.bias"] = torch.zeros(cout)
    elif cin != cout:                                            # :66-69
        st[f"{p}.conv_skip.0.weight"] = torch.zeros(cout, cin, 1, 1)
        st[f"{p}.conv_skip.0.bias"] = torch.zeros(cout)
        _bn_entries(f"{p}.conv_skip.1", cout, st)


def make_res_unet_state(in_channels: int = 1, out_channels: int = 1, res_type: str = "next",
                        channel_mults: Sequence[int] = (1, 2, 4, 8, 8, 8, 8, 8)) -> OrderedDict:
    """Keys/shapes of ``ResUnet(in_channels, out_channels, res_type, channel_mults).state_dict()`` (:258-315)."""
    st: OrderedDict[str, torch.Tensor] = OrderedDict()
    st["in_conv.weight"] = torch.zeros(64, in_channels, 3, 3)
    st["in_conv.bias"] = torch.zeros(64)
    cin = 64
    L = len(channel_mults)
    for level, mult in enumerate(channel_mults):
        _block_state(st, f"encoders.{level}.encode.0", res_type, cin, mult * 64)
        cin = mult * 64
    j = 0
    for level in reversed(range(L - 1)):
        c = channel_mults[level] * 64
        _block_state(st, f"decoders.{j}.decode.0", res_type, cin, c)
        cin = c * 2
        j += 1
    _block_state(st, f"decoders.{j}.decode.0", res_type, cin, channel_mults[0] * 64)
    st["out.0.weight"] = torch.zeros(out_channels, channel_mults[0] * 64, 3, 3)
    st["out.0.bias"] = torch.zeros(out_channels)
    return st


def res_type_of(st) -> str:
    if "encoders.0.encode.0.conv_block.0.running_mean" in st:
        return "v2"
    w3 = st.get("encoders.0.encode.0.conv_block.3.weight")
    if w3 is not None and w3.shape[2] == 3 and w3.shape[1] != w3.shape[0]:
        return "next"
    return "50" if "encoders.0.encode.0.conv_block.6.weight" in st else "18"


def _conv(st, key, h):
    w = st[key + ".weight"]
    return F.conv2d(h, w, st[key + ".bias"], padding=w.shape[2] // 2, groups=h.shape[1] // w.shape[1])


def residual_block(st, p, x, res_type, training):
    """conv_block(x) + conv_skip(x), ReLU behind the sum for "18"/"50" (models/res_unet.py:73-74,104-105,170-171)."""
    layout, post = BLOCKS[res_type]
    h = x
    for idx, kind in enumerate(layout):
        q = f"{p}.conv_block.{idx}"
        if kind == "c":
            h = _conv(st, q, h)
        elif kind == "b":
            h = _bn(st, q, h, training)
        else:
            h = F.relu(h)
    if (p + ".conv_skip.2.weight") in st:                       # v2: BatchNorm -> ReLU -> 1x1 conv
        s = _conv(st, p + ".conv_skip.2", F.relu(_bn(st, p + ".conv_skip.0", x, training)))
    elif (p + ".conv_skip.0.weight") in st:
        s = _bn(st, p + ".conv_skip.1", _conv(st, p + ".conv_skip.0", x), training)
    else:
        s = x
    h = h + s
    return F.relu(h) if post else h


def res_unet_forward(st: OrderedDict, x: torch.Tensor, training: bool = True, return_feats: bool = False,
                     dropout: float = 0.0, mask_log=None, masks=None):
    """``ResUnet.forward`` (models/res_unet.py:317-334): in_conv, encoders = block -> MaxPool2d(2), decoders =
    block -> Dropout2d -> nearest Upsample(2) with ``cat([h, skips.pop()])`` in front of every decoder but the
    first, out = Conv2d 3x3 -> Tanh."""
    rt = res_type_of(st)
    L = sum(1 for k in st if k.startswith("encoders.") and k.endswith("encode.0.conv_block.0.weight"))
    last_conv = {"18": 3, "50": 6, "next": 6, "v2": 5}[rt]
    mults = [st[f"encoders.{i}.encode.0.conv_block.{last_conv}.weight"].shape[0] // 64 for i in range(L)]
    h = x if x.dtype == torch.float64 else x.to(torch.float32)
    h = F.conv2d(h, st["in_conv.weight"], st["in_conv.bias"], padding=1)
    acts = {"in": h}
    skips = []
    for i in range(L):
        h = residual_block(st, f"encoders.{i}.encode.0", h, rt, training)
        h = F.max_pool2d(h, 2)
        skips.append(h)
        acts[f"enc{i}"] = h
    skips.pop()
    for j in range(L):
        if j != 0:
            h = torch.cat([h, skips.pop()], dim=1)
        h = residual_block(st, f"decoders.{j}.decode.0", h, rt, training)
        level = L - 2 - j
        p = dropout if (j < L - 1 and mults[level] == max(mults) and level > L - 5) else 0.0   # :286-289
        if training and p > 0:
            h = apply_dropout2d(h, p, j, mask_log, masks)
        h = F.interpolate(h, scale_factor=2)                      # nn.Upsample default: nearest
        acts[f"dec{j}"] = h
    out = torch.tanh(F.conv2d(h, st["out.0.weight"], st["out.0.bias"], padding=1))
    return (out, acts) if return_feats else out
