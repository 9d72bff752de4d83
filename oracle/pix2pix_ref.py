"""Oracle: functional CPU restatement of the reference U-Net generator and PatchGAN.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it restates (paths relative to /root/reference).

State is carried as an ``OrderedDict[str, torch.Tensor]`` whose keys and
logical shapes are exactly the reference modules' ``state_dict()`` keys
(relative to the ``Unet`` / ``Discriminator`` module), so a reference state
dict can be dropped in unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default, models/pix2pix.py:70,106
BN_MOMENTUM = 0.1    # torch.nn.BatchNorm2d default


# --------------------------------------------------------------------------
# state construction
# --------------------------------------------------------------------------
def _bn_entries(prefix: str, c: int, st: OrderedDict):
    st[prefix + ".weight"] = torch.ones(c)
    st[prefix + ".bias"] = torch.zeros(c)
    st[prefix + ".running_mean"] = torch.zeros(c)
    st[prefix + ".running_var"] = torch.ones(c)
    st[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def make_unet_state(in_channels: int = 1, out_channels: int = 1,
                    channel_mults: Sequence[int] = (1, 2, 4, 8, 8, 8, 8, 8)) -> OrderedDict:
    """Zero-filled state with the keys/shapes of ``Unet.state_dict()``.

    Topology follows models/pix2pix.py:130-196: encoders[0] is a bare Conv2d
    (:141-147), encoders[1..L-1] are EncoderBlocks whose norm is disabled on the
    last level (:157), decoders[0..L-2] are DecoderBlocks (:167-183) and the
    last decoder is a bare ConvTranspose2d (:185-193).
    ConvTranspose2d weights are [Cin, Cout, 4, 4].
    """
    st: OrderedDict[str, torch.Tensor] = OrderedDict()
    L = len(channel_mults)
    c = channel_mults[0] * 64
    st["encoders.0.weight"] = torch.zeros(c, in_channels, 4, 4)
    st["encoders.0.bias"] = torch.zeros(c)
    cin = c
    for level in range(1, L):
        c = channel_mults[level] * 64
        st[f"encoders.{level}.encode.1.weight"] = torch.zeros(c, cin, 4, 4)
        st[f"encoders.{level}.encode.1.bias"] = torch.zeros(c)
        if level != L - 1:
            _bn_entries(f"encoders.{level}.encode.2", c, st)
        cin = c
    j = 0
    for level in reversed(range(L - 1)):
        c = channel_mults[level] * 64
        st[f"decoders.{j}.decode.1.weight"] = torch.zeros(cin, c, 4, 4)
        st[f"decoders.{j}.decode.1.bias"] = torch.zeros(c)
        _bn_entries(f"decoders.{j}.decode.2", c, st)
        cin = c * 2
        j += 1
    st[f"decoders.{j}.weight"] = torch.zeros(cin, out_channels, 4, 4)
    st[f"decoders.{j}.bias"] = torch.zeros(out_channels)
    return st


def make_disc_state(in_channels: int = 1) -> OrderedDict:
    """Keys/shapes of ``Discriminator(in_channels).state_dict()``
    (models/wrapper.py:225-234).  ``norm`` is never enabled (wrapper.py:192,
    229-232) so there are no InstanceNorm entries."""
    st: OrderedDict[str, torch.Tensor] = OrderedDict()
    chans = [in_channels * 2, 64, 128, 256, 512]
    for i in range(4):
        st[f"discriminator.{i}.block.0.weight"] = torch.zeros(chans[i + 1], chans[i], 4, 4)
        st[f"discriminator.{i}.block.0.bias"] = torch.zeros(chans[i + 1])
    st["discriminator.4.weight"] = torch.zeros(1, 512, 4, 4)
    return st


def init_state_portable(st: OrderedDict, seed: int, perturb_bn: bool = False) -> OrderedDict:
    """Fill a state dict from a *portable* generator (numpy PCG64), so that the
    same weights can be regenerated on any box without shipping them.

    Mirrors ``init_weights`` (models/utils.py:15-28): conv / conv-transpose
    weights ~ N(0, 0.02); norm affine = (1, 0).  Biases keep torch's default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) *distribution* (wrapper.py:37 only
    re-initialises weights) but are drawn from the portable stream.
    ``perturb_bn`` additionally jitters gamma/beta/running stats so that tests
    exercise non-trivial affine parameters.
    """
    rng = np.random.default_rng(seed)
    for k, v in st.items():
        if k.endswith("num_batches_tracked"):
            v.zero_()
        elif v.dim() == 4:
            v.copy_(torch.from_numpy((rng.standard_normal(v.shape) * 0.02).astype(np.float32)))
        elif k.endswith("running_mean"):
            v.zero_()
            if perturb_bn:
                v.copy_(torch.from_numpy((rng.standard_normal(v.shape) * 0.05).astype(np.float32)))
        elif k.endswith("running_var"):
            v.fill_(1.0)
            if perturb_bn:
                v.copy_(torch.from_numpy((1.0 + 0.2 * rng.random(v.shape)).astype(np.float32)))
        elif (k.rsplit(".", 1)[0] + ".running_mean") in st:   # affine parameters of a BatchNorm
            if k.endswith("weight"):
                v.fill_(1.0)
                if perturb_bn:
                    v.copy_(torch.from_numpy((1.0 + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)))
            else:
                v.zero_()
                if perturb_bn:
                    v.copy_(torch.from_numpy((0.1 * rng.standard_normal(v.shape)).astype(np.float32)))
        else:  # conv / conv-transpose bias
            wkey = k[:-len("bias")] + "weight"
            w = st[wkey]
            # torch: fan_in = weight.size(1) * receptive field, also for ConvTranspose2d
            fan_in = w.shape[1] * w.shape[2] * w.shape[3]
            bound = 1.0 / np.sqrt(fan_in)
            v.copy_(torch.from_numpy(rng.uniform(-bound, bound, v.shape).astype(np.float32)))
    return st


# --------------------------------------------------------------------------
# forward passes
# --------------------------------------------------------------------------
def _bn(st, prefix, h, training: bool):
    """nn.BatchNorm2d (models/pix2pix.py:70,106): train = batch statistics with
    biased variance for normalisation, running stats updated with momentum 0.1
    and the unbiased variance; eval = running statistics."""
    if training:
        st[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(
        h, st[prefix + ".running_mean"], st[prefix + ".running_var"],
        st[prefix + ".weight"], st[prefix + ".bias"],
        training=training, momentum=BN_MOMENTUM, eps=BN_EPS)


def dropout_rates(st: OrderedDict, dropout: float):
    """Dropout2d rate of every DecoderBlock: `dropout` for the blocks at the widest channel multiple and
    level > L - 5, else 0 (models/pix2pix.py:176-179, models/attention_unet.py:166-171)."""
    L = 1 + sum(1 for k in st if k.startswith("encoders.") and k.endswith("encode.1.weight"))
    mults = [st["encoders.0.weight"].shape[0] // 64] + \
            [st[f"encoders.{i}.encode.1.weight"].shape[0] // 64 for i in range(1, L)]
    out = []
    for j, level in enumerate(reversed(range(L - 1))):
        out.append(dropout if (mults[level] == max(mults) and level > L - 5) else 0.0)
    return out


def apply_dropout2d(h, p, j, mask_log=None, masks=None):
    """nn.Dropout2d(p) in training mode (models/pix2pix.py:108).  The mask is drawn exactly as torch's
    feature dropout draws it (a Bernoulli(1-p) sample per (n, c), scaled by 1/(1-p), from the global CPU
    generator), so a run seeded like the reference sees the reference's masks; `masks` replays given ones."""
    if masks is not None:
        m = masks.pop(0)
    else:
        m = F.dropout2d(torch.ones(h.shape[0], h.shape[1], 1, 1, dtype=h.dtype), p, training=True)
    if mask_log is not None:
        mask_log.append((j, m.clone()))
    return h * m


def unet_forward(st: OrderedDict, x: torch.Tensor, training: bool = True,
                 return_feats: bool = False, dropout: float = 0.0, mask_log=None, masks=None):
    """``Unet.forward`` (models/pix2pix.py:198-216).

    Encoder i>=1 = LeakyReLU(0.2) -> Conv2d(k4,s2,p1) -> BN (pix2pix.py:61-71);
    the stored skip is the *pre-activation* block output.  Decoder j<L-1 =
    ReLU -> ConvTranspose2d(k4,s2,p1) -> BN (:97-108); the last decoder is a bare
    ConvTranspose2d with NO ReLU in front (:185-193); output = tanh (:216).
    Skip concat puts the decoder output first (:212).  dropout must be 0.
    """
    L = 1 + sum(1 for k in st if k.startswith("encoders.") and k.endswith("encode.1.weight"))
    h = x if x.dtype == torch.float64 else x.to(torch.float32)    # :199 (fp64 only for noise-floor studies)
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
    feats.pop()                                                    # :208
    drops = dropout_rates(st, dropout)
    for j in range(L - 1):
        p = f"decoders.{j}.decode"
        if j != 0:
            h = torch.cat([h, feats.pop()], dim=1)                 # :212
        h = F.relu(h)
        h = F.conv_transpose2d(h, st[p + ".1.weight"], st[p + ".1.bias"], stride=2, padding=1)
        h = _bn(st, p + ".2", h, training)
        if training and drops[j] > 0:
            h = apply_dropout2d(h, drops[j], j, mask_log, masks)
        acts[f"dec{j}"] = h
    j = L - 1
    if j != 0:
        h = torch.cat([h, feats.pop()], dim=1)
    h = F.conv_transpose2d(h, st[f"decoders.{j}.weight"], st[f"decoders.{j}.bias"],
                           stride=2, padding=1)
    out = torch.tanh(h)                                            # :216
    if return_feats:
        return out, acts
    return out


def disc_forward(st: OrderedDict, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``Discriminator.forward`` (models/wrapper.py:236-238): cat([x, y]) ->
    4 x [Conv2d(k4,s2,p1) -> Identity -> LeakyReLU(0.2)] (:196-206, act AFTER the
    conv) -> Conv2d(512,1,k4,s1,p1,bias=False) (:233)."""
    h = torch.cat([x, y], dim=1)
    for i in range(4):
        p = f"discriminator.{i}.block.0"
        h = F.conv2d(h, st[p + ".weight"], st[p + ".bias"], stride=2, padding=1)
        h = F.leaky_relu(h, 0.2)
    return F.conv2d(h, st["discriminator.4.weight"], None, stride=1, padding=1)
