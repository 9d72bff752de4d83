"""Tensor fingerprints used by the golden fixtures (TEST INFRASTRUCTURE ONLY).

A fingerprint is a small float64 vector: [l2, sum, abs-mean, max-abs] followed by
32 samples taken at a fixed stride over the flattened tensor, so that a fixture
can pin a multi-megabyte tensor in a few hundred bytes.
"""
import numpy as np
import torch

N_SAMPLES = 32


def fingerprint(t) -> np.ndarray:
    a = t.detach().cpu().to(torch.float64).reshape(-1).numpy() if torch.is_tensor(t) \
        else np.asarray(t, dtype=np.float64).reshape(-1)
    n = a.size
    idx = (np.arange(N_SAMPLES, dtype=np.int64) * max(n // N_SAMPLES, 1) + (n // (2 * N_SAMPLES))) % n
    head = np.array([np.sqrt((a * a).sum()), a.sum(), np.abs(a).mean(), np.abs(a).max()])
    return np.concatenate([head, a[idx]])


def fingerprint_close(got: np.ndarray, want: np.ndarray, rtol: float, atol_scale: float = 1.0):
    """Relative check: every entry within rtol of the tensor's scale.
    The scale for the samples is the tensor's max-abs (want[3]); for l2/sum it is l2."""
    l2, mx = want[0], want[3]
    errs = np.abs(got - want)
    tol = np.empty_like(want)
    tol[0] = rtol * max(l2, 1e-30)
    # sum can cancel: bound it by rtol * l1 (= abs-mean * n is unknown) -> use l2 * 32 as a loose scale
    tol[1] = rtol * max(l2, 1e-30) * 64
    tol[2] = rtol * max(want[2], 1e-30)
    tol[3] = rtol * max(mx, 1e-30) * 4
    tol[4:] = rtol * max(mx, 1e-30) * 4
    tol *= atol_scale
    bad = errs > tol
    return (not bad.any()), float((errs / np.maximum(tol, 1e-300)).max())
