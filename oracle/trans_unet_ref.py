"""Oracle: functional CPU restatement of the reference TransUNet.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates ``TransUnet`` and its blocks (models/trans_unet.py:35-255;
paths relative to /root/reference) with torch-CPU ops and an explicit multi-head attention.  State keys / shapes are
those of ``TransUnet.state_dict()``.  The transformer layers are ``nn.TransformerEncoderLayer`` defaults of the pinned
torch 2.0.0: post-norm, dim_feedforward 2048, LayerNorm eps 1e-5, erf GELU, ``batch_first=False`` -- the module is fed
``[n, patches, dim]``, so the attention sequence is the image batch.  Dropout inside the transformer
(p > 0) is restated in distribution, with the oracle's own draws (see ``encoder_layer``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .pix2pix_ref import _bn, _bn_entries

LAYERS = 12        # models/trans_unet.py:85
HEADS = 8          # :25
FF = 2048          # nn.TransformerEncoderLayer default dim_feedforward
IMAGE = 256        # :22
# False: the attention is written out below (the restatement).  True: torch's own F.multi_head_attention_forward, the
# operator the reference's nn.MultiheadAttention dispatches to -- bit-identical to the recorded fixtures, used by
# tests/test_oracle_golden.py to pin everything around the attention at 2e-5 and the written-out form against it.
USE_ATEN_MHA = False


def make_trans_unet_state(in_channels: int = 1, out_channels: int = 1, channel_mults: Sequence[int] = (1, 2, 2, 4, 4),
                          patch_size: int = 2) -> OrderedDict:
    """Keys/shapes of ``TransUnet(in, out, 256, channel_mults, patch_size, 8, dropout).state_dict()`` (:52-99)."""
    st: OrderedDict[str, torch.Tensor] = OrderedDict()
    st["in_conv.weight"] = torch.zeros(64, in_channels, 3, 3)
    st["in_conv.bias"] = torch.zeros(64)
    cin = 64
    for i, mult in enumerate(channel_mults):                      # EncoderBlock :198-236
        c, b, p = mult * 64, cin // 4, f"encoders.{i}"
        st[f"{p}.decode.0.weight"] = torch.zeros(b, cin, 1, 1)
        _bn_entries(f"{p}.decode.1", b, st)
        st[f"{p}.decode.3.weight"] = torch.zeros(b, b, 3, 3)
        _bn_entries(f"{p}.decode.4", b, st)
        st[f"{p}.decode.6.weight"] = torch.zeros(c, b, 1, 1)
        _bn_entries(f"{p}.decode.7", c, st)
        st[f"{p}.skip.0.weight"] = torch.zeros(c, cin, 1, 1)
        _bn_entries(f"{p}.skip.1", c, st)
        cin = c
    size = IMAGE // (2 ** len(channel_mults))
    D = cin * patch_size * patch_size                             # :132
    P = (size ** 2) // (patch_size ** 2)                          # :133
    v = "vit_bottleneck"
    st[f"{v}.pos_embedding"] = torch.zeros(1, P, D)
    e = f"{v}.to_patch_embedding"                                 # index 0 is the parameter-free Rearrange
    st[f"{e}.1.weight"], st[f"{e}.1.bias"] = torch.ones(D), torch.zeros(D)
    st[f"{e}.2.weight"], st[f"{e}.2.bias"] = torch.zeros(D, D), torch.zeros(D)
    st[f"{e}.3.weight"], st[f"{e}.3.bias"] = torch.ones(D), torch.zeros(D)
    for li in range(LAYERS):
        q = f"{v}.transformer.layers.{li}"
        st[f"{q}.self_attn.in_proj_weight"] = torch.zeros(3 * D, D)
        st[f"{q}.self_attn.in_proj_bias"] = torch.zeros(3 * D)
        st[f"{q}.self_attn.out_proj.weight"] = torch.zeros(D, D)
        st[f"{q}.self_attn.out_proj.bias"] = torch.zeros(D)
        st[f"{q}.linear1.weight"] = torch.zeros(FF, D)
        st[f"{q}.linear1.bias"] = torch.zeros(FF)
        st[f"{q}.linear2.weight"] = torch.zeros(D, FF)
        st[f"{q}.linear2.bias"] = torch.zeros(D)
        for nm in ("norm1", "norm2"):
            st[f"{q}.{nm}.weight"] = torch.ones(D)
            st[f"{q}.{nm}.bias"] = torch.zeros(D)
    j = 0
    for mult in reversed(list(channel_mults[:-1])):               # :89-96
        c = mult * 64
        _decoder_state(st, f"decoders.{j}", cin, c)
        cin, j = c * 2, j + 1
    _decoder_state(st, f"decoders.{j}", cin, 64)
    st["out.0.weight"] = torch.zeros(out_channels, 64, 3, 3)
    st["out.0.bias"] = torch.zeros(out_channels)
    return st


def _decoder_state(st, p, cin, c):
    st[f"{p}.decode.0.weight"] = torch.zeros(c, cin, 3, 3)
    st[f"{p}.decode.0.bias"] = torch.zeros(c)
    _bn_entries(f"{p}.decode.1", c, st)
    st[f"{p}.decode.3.weight"] = torch.zeros(c, c, 3, 3)
    st[f"{p}.decode.3.bias"] = torch.zeros(c)
    _bn_entries(f"{p}.decode.4", c, st)


def init_trans_state_portable(st: OrderedDict, seed: int, perturb: bool = True) -> OrderedDict:
    """Portable (numpy PCG64) fill in the spirit of ``init_weights`` (models/utils.py:15-28): conv / linear weights
    ~ N(0, 0.02), ``pos_embedding`` ~ N(0, 1) (:147), the attention in-projection Xavier-uniform (torch default; it is a
    bare Parameter that ``init_weights`` does not reach), norm affines (1, 0) -- jittered with ``perturb`` so that
    tests see non-trivial affines, like ``init_state_portable(perturb_bn=True)`` -- and small random biases."""
    rng = np.random.default_rng(seed)

    def normal(v, std, mean=0.0):
        v.copy_(torch.from_numpy((mean + std * rng.standard_normal(tuple(v.shape))).astype(np.float32)))

    def uniform(v, bound):
        v.copy_(torch.from_numpy(rng.uniform(-bound, bound, tuple(v.shape)).astype(np.float32)))

    for k, v in st.items():
        stem = k.rsplit(".", 1)[0]
        if k.endswith("num_batches_tracked"):
            v.zero_()
        elif k.endswith("pos_embedding"):
            normal(v, 1.0)
        elif k.endswith("in_proj_weight"):
            uniform(v, math.sqrt(6.0 / (v.shape[0] + v.shape[1])))
        elif k.endswith("in_proj_bias"):
            normal(v, 0.02 if perturb else 0.0)
        elif v.dim() in (2, 4):
            normal(v, 0.02)
        elif k.endswith("running_mean"):
            normal(v, 0.05 if perturb else 0.0)
        elif k.endswith("running_var"):
            v.copy_(torch.from_numpy((1.0 + (0.2 if perturb else 0.0) * rng.random(tuple(v.shape))).astype(np.float32)))
        elif (stem + ".running_mean") in st or "norm" in stem or "to_patch_embedding.1" in stem or "to_patch_embedding.3" in stem:
            if k.endswith("weight"):
                normal(v, 0.1 if perturb else 0.0, 1.0)
            else:
                normal(v, 0.1 if perturb else 0.0)
        else:                                                     # conv / linear bias: torch's U(-1/sqrt(fan_in), ...)
            w = st[stem + ".weight"]
            uniform(v, 1.0 / math.sqrt(w[0].numel()))
    return st


def _conv(st, key, h, stride=1):
    w = st[key + ".weight"]
    return F.conv2d(h, w, st.get(key + ".bias"), stride=stride, padding=w.shape[2] // 2)


def encoder_block(st, p, x, training):
    """``EncoderBlock.forward`` (:235-236): ReLU(decode(x) + skip(x))."""
    h = F.relu(_bn(st, p + ".decode.1", _conv(st, p + ".decode.0", x), training))
    h = F.relu(_bn(st, p + ".decode.4", _conv(st, p + ".decode.3", h, stride=2), training))
    h = _bn(st, p + ".decode.7", _conv(st, p + ".decode.6", h), training)
    s = _bn(st, p + ".skip.1", _conv(st, p + ".skip.0", x, stride=2), training)
    return F.relu(h + s)


def decoder_block(st, p, x, training):
    """``DecoderBlock.forward`` (:249-258)."""
    h = F.relu(_bn(st, p + ".decode.1", _conv(st, p + ".decode.0", x), training))
    h = F.relu(_bn(st, p + ".decode.4", _conv(st, p + ".decode.3", h), training))
    return F.interpolate(h, scale_factor=2)                       # nn.Upsample default: nearest


def _ln(st, key, t):
    return F.layer_norm(t, (t.shape[-1],), st[key + ".weight"], st[key + ".bias"], 1e-5)


def _drop(x, p, site, li, mask_log):
    """nn.Dropout in training mode, logged for replay.  The mask is what ``F.dropout`` draws for a tensor of this size
    from the global CPU generator (ATen: ``empty_like(x).bernoulli_(1 - p).div_(1 - p)``), obtained by dropping out a
    tensor of ones -- so that, from the same generator state, the oracle consumes the generator exactly as the
    reference's ``nn.Dropout`` modules and the attention operator's internal ``at::dropout`` do (pinned by the
    ``ref_trans4_gan_dropout`` fixture, tests/test_oracle_golden.py)."""
    if not p:
        return x
    m = F.dropout(torch.ones(x.shape, dtype=x.dtype), p, True)
    if mask_log is not None:
        mask_log.append((site, li, m))
    return x * m


def encoder_layer(st, q, t, p=0.0, li=0, mask_log=None):
    """One post-norm ``nn.TransformerEncoderLayer`` on t = [S, B, E] (sequence first).  ``p`` > 0 (training mode): its
    four Dropout sites -- attention weights, dropout1 behind the attention block, dropout inside and dropout2 behind the
    feed-forward block, drawn in that order.  The attention dropout of the reference happens inside the ATen attention
    operator (``at::dropout`` on the [B, H, S, S] weights of the math path); the written-out branch draws a mask of the
    same size at the same point of the generator stream."""
    S, B, E = t.shape
    hd = E // HEADS
    if USE_ATEN_MHA:
        o, _ = F.multi_head_attention_forward(
            t, t, t, E, HEADS, st[q + ".self_attn.in_proj_weight"], st[q + ".self_attn.in_proj_bias"], None, None, False,
            float(p), st[q + ".self_attn.out_proj.weight"], st[q + ".self_attn.out_proj.bias"], training=True,
            need_weights=False)
    else:
        qkv = F.linear(t, st[q + ".self_attn.in_proj_weight"], st[q + ".self_attn.in_proj_bias"])
        qq, kk, vv = (c.reshape(S, B * HEADS, hd).transpose(0, 1) for c in qkv.chunk(3, dim=-1))   # [B*H, S, hd]
        att = _drop(torch.softmax(qq @ kk.transpose(1, 2) / math.sqrt(hd), dim=-1), p, "attn", li, mask_log)
        o = (att @ vv).transpose(0, 1).reshape(S, B, E)
        o = F.linear(o, st[q + ".self_attn.out_proj.weight"], st[q + ".self_attn.out_proj.bias"])
    t = _ln(st, q + ".norm1", t + _drop(o, p, "sa", li, mask_log))
    f = _drop(F.gelu(F.linear(t, st[q + ".linear1.weight"], st[q + ".linear1.bias"])), p, "ff", li, mask_log)
    f = _drop(F.linear(f, st[q + ".linear2.weight"], st[q + ".linear2.bias"]), p, "out", li, mask_log)
    return _ln(st, q + ".norm2", t + f)


def vit_bottleneck(st, h, acts=None, drop=0.0, mask_log=None):
    """``VisionTransformer.forward`` (:170-175)."""
    v = "vit_bottleneck"
    n, c, hs, ws = h.shape
    _, P, D = st[v + ".pos_embedding"].shape
    p = int(round(math.sqrt(D // c)))
    g = hs // p
    # Rearrange "n c (h p1) (w p2) -> n (h w) (p1 p2 c)"
    t = h.reshape(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)
    t = _ln(st, v + ".to_patch_embedding.1", t)
    t = F.linear(t, st[v + ".to_patch_embedding.2.weight"], st[v + ".to_patch_embedding.2.bias"])
    t = _ln(st, v + ".to_patch_embedding.3", t)
    t = t + st[v + ".pos_embedding"]
    for li in range(LAYERS):
        t = encoder_layer(st, f"{v}.transformer.layers.{li}", t, drop, li, mask_log)  # [n, P, D] read as [S = n, B = P, E]
        if acts is not None:
            acts[f"vit{li}"] = t
    # Rearrange "n (h w) (p1 p2 c) -> n c (h p1) (w p2)"
    return t.reshape(n, g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(n, c, hs, ws)


def trans_unet_forward(st: OrderedDict, x: torch.Tensor, training: bool = True, return_feats: bool = False,
                       dropout: float = 0.0, mask_log=None, **_):
    """``TransUnet.forward`` (:101-117).  ``dropout``: rate of the transformer layers' Dropout (training mode only)."""
    L = sum(1 for k in st if k.startswith("encoders.") and k.endswith(".decode.0.weight"))
    h = x if x.dtype == torch.float64 else x.to(torch.float32)
    h = F.conv2d(h, st["in_conv.weight"], st["in_conv.bias"], padding=1)
    acts = {}
    skips = []
    for i in range(L):
        h = encoder_block(st, f"encoders.{i}", h, training)
        skips.append(h)
        acts[f"enc{i}"] = h
    skips.pop()
    h = vit_bottleneck(st, h, acts, dropout if training else 0.0, mask_log)
    acts["vit"] = h
    for j in range(L):
        if j != 0:
            h = torch.cat([h, skips.pop()], dim=1)
        h = decoder_block(st, f"decoders.{j}", h, training)
        acts[f"dec{j}"] = h
    out = torch.tanh(F.conv2d(h, st["out.0.weight"], st["out.0.bias"], padding=1))
    return (out, acts) if return_feats else out
