"""Pointwise (1 x 1) convolutions of the residual U-Net (BASELINE configs[3]) alone on the chip: forward (with BatchNorm
partial statistics) and input gradient, the streaming kernel (pwx_k, gg_pw.hip) against the tile kernel (pwx=0).
    python scripts/bench_pw.py [name=value ...]          (GPU box; extra tunables apply to both)"""
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


# N, H, C1, C2, Cout
SHAPES = [(16, 512, 64, 0, 128), (16, 512, 128, 0, 64), (16, 256, 64, 0, 128), (16, 256, 128, 0, 128), (16, 256, 64, 64, 128),
          (16, 256, 128, 0, 64), (16, 256, 64, 64, 64), (16, 128, 128, 128, 64), (16, 128, 128, 0, 256), (16, 128, 128, 128, 128)]
for N, H, C1, C2, K in SHAPES:
    Cin, M = C1 + C2, N * H * H
    d = ops.make_desc(dt, 0, N, H, H, C1, C2, K, 1, 0, 0, ops.ACT_NONE, kernel=1)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev)
    x1 = torch.randn(M, C1, device=dev).to(dt)
    x2 = torch.randn(M, C2, device=dev).to(dt) if C2 else None
    w = torch.randn(K * Cin, device=dev).to(dt)
    bias = torch.randn(K, device=dev)
    y = torch.empty(M, K, dtype=dt, device=dev)
    dy = torch.randn(M, K, device=dev).to(dt)
    dx1 = torch.empty(M, C1, dtype=dt, device=dev)
    dx2 = torch.empty(M, C2, dtype=dt, device=dev) if C2 else None
    stats = torch.empty(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * K, device=dev)
    byt = M * (Cin + K) * 2
    line = f"{N}x{H}x{H} {C1}+{C2}->{K}: {byt / 1e6:7.1f} MB |"
    for pwx in (1, 0):
        ops.set_tunable("pwx", pwx)
        tf = timeit(lambda: ops.conv_fwd(d, x1, x2, w, bias, y_raw=y, stats=stats))
        td = timeit(lambda: ops.conv_dgrad(d, dy, w, dx1, dx2))
        line += f" {ops.conv_kernel_name(d, 0):>42s} f {tf:6.1f} us {byt / tf / 1e6:5.2f} TB/s  d {td:6.1f} us {byt / td / 1e6:5.2f} TB/s |"
    ops.set_tunable("pwx")
    print(line, flush=True)
