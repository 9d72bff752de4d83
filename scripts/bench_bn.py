"""Bandwidth of the BatchNorm / activation backward kernels at the bench model's shapes (bs 64, 256x256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pai_bootstrap
pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops

dev = torch.device("cuda", 0)
dt = torch.bfloat16
SHAPES = [("dec6", 1 << 20, 64, False), ("dec5", 1 << 18, 128, False), ("enc1", 1 << 18, 128, True),
          ("enc2", 1 << 16, 256, True), ("dec4", 1 << 16, 256, False), ("enc3", 1 << 14, 512, True),
          ("dec2", 1 << 12, 512, False), ("enc5", 1 << 10, 512, True)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, C, skip in SHAPES:
    g1 = torch.randn(M, C, device=dev).to(dt)
    g2 = torch.randn(M, C, device=dev).to(dt) if skip else None
    a = torch.randn(M, C, device=dev).to(dt)
    z = torch.randn(M, C, device=dev).to(dt)
    du = torch.empty_like(z)
    dz = torch.empty_like(z)
    mean = torch.zeros(C, device=dev)
    rstd = torch.ones(C, device=dev)
    gamma = torch.ones(C, device=dev)
    partials = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * C, device=dev)
    sums = torch.empty(2 * C, device=dev)
    dg = torch.zeros(C, device=dev)
    db = torch.zeros(C, device=dev)
    T = M * C * 2
    t_r = timeit(lambda: ops.bn_bwd_reduce(dt, g1, 1, g2, 2 if skip else 0, a, z, M, C, mean, rstd, du, partials, sums, dg, db))
    t_a = timeit(lambda: ops.bn_bwd_apply(dt, du, z, M, C, mean, rstd, gamma, sums, dz))
    t_c = timeit(lambda: ops.act_bwd(dt, g1, 1, g2, 2 if skip else 0, a, M * C, du))
    nr = 5 if skip else 4
    print(f"{name:5s} M={M:8d} C={C:4d}  reduce+fin {t_r:7.1f} us {nr*T/t_r/1e6:6.2f} TB/s | apply {t_a:7.1f} us {3*T/t_a/1e6:6.2f} TB/s"
          f" | act_bwd {t_c:7.1f} us {(nr-1)*T/t_c/1e6:6.2f} TB/s")
