"""BatchNorm-backward passes at the residual U-Net's (BASELINE configs[3]) tensor sizes: python scripts/bench_bn_big.py
[name=value ...]   (tunables, e.g. bn_reduce_chunk=0 for one slab of rows per block)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pai_bootstrap

pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ops.set_tunable(k, int(v))
dev, dt = torch.device("cuda", 0), torch.bfloat16


def timeit(fn, n=10):
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


for M, C in ((16 * 512 * 512, 128), (16 * 512 * 512, 64), (16 * 256 * 256, 128), (16 * 128 * 128, 256), (16 * 64 * 64, 512)):
    g = torch.randn(M, C, device=dev).to(dt)
    z = torch.randn(M, C, device=dev).to(dt)
    dz = torch.empty_like(z)
    f32 = dict(dtype=torch.float32, device=dev)
    mean, rstd, gamma = torch.zeros(C, **f32), torch.ones(C, **f32), torch.ones(C, **f32)
    scale, shift = torch.ones(C, **f32), torch.zeros(C, **f32)
    part = torch.empty(ops.bn_bwd_partial_rows(M) * 2 * C, **f32)
    sums = torch.empty(2 * C, **f32)
    T = M * C * 2
    t_r = timeit(lambda: ops.bn_bwd_reduce_affine(dt, g, 2, None, 0, z, M, C, scale, shift, mean, rstd, None, part, sums, None, None))
    t_a = timeit(lambda: ops.bn_bwd_apply_affine(dt, g, 2, z, M, C, scale, shift, mean, rstd, gamma, sums, dz))
    print(f"M={M:8d} C={C:4d}  reduce+finalize {t_r:7.1f} us {2 * T / t_r / 1e6:5.2f} TB/s | apply {t_a:7.1f} us {3 * T / t_a / 1e6:5.2f} TB/s",
          flush=True)
