"""Helpers for the -m gpu parity tests: NCHW <-> NHWC marshalling and C-ABI conv calls."""
import numpy as np
import torch


def dev():
    return torch.device("cuda:0")


def nhwc(t: torch.Tensor, dtype) -> torch.Tensor:
    """NCHW cpu fp32 -> flat NHWC device tensor of the storage dtype."""
    return t.permute(0, 2, 3, 1).contiguous().to(dev()).to(dtype).reshape(-1)


def from_nhwc(flat: torch.Tensor, n, h, w, c) -> torch.Tensor:
    return flat.float().cpu().view(n, h, w, c).permute(0, 3, 1, 2).contiguous()


def fwd_pack(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """torch weight -> fp32 [Cout][kh][kw][Cin] on device."""
    p = w.permute(1, 2, 3, 0) if transposed else w.permute(0, 2, 3, 1)
    return p.contiguous().to(dev()).float()


def unpack_fwd(flat: torch.Tensor, cout, cin, transposed: bool) -> torch.Tensor:
    p = flat.float().cpu().view(cout, 4, 4, cin)
    return (p.permute(3, 0, 1, 2) if transposed else p.permute(0, 3, 1, 2)).contiguous()


def rnd(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def q(t, dtype):
    """Round a cpu fp32 tensor through the storage dtype (so the CPU reference sees the same inputs)."""
    return t.to(dtype).float()


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    return float((got.double() - want.double()).norm() / max(float(want.double().norm()), 1e-30))


def max_err(got, want) -> float:
    return float((got.double() - want.double()).abs().max() / max(float(want.double().abs().max()), 1e-30))
