"""Micro-benchmark of the small-channel / grouped convolutions (gg_small.hip) against the HBM time of their tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
# name, k, N, H, Cin, Cout, groups
LAYERS = [("rnx 3x3 g32 512", 3, 16, 512, 128, 128, 32), ("rnx 3x3 g32 256", 3, 16, 256, 128, 128, 32),
          ("rnx 3x3 dense 512", 3, 16, 512, 128, 128, 1), ("rnx 1x1 64>128 512", 1, 16, 512, 64, 128, 1),
          ("rnx 1x1 128>64 512", 1, 16, 512, 128, 64, 1), ("rnx 1x1 128>128 512", 1, 16, 512, 128, 128, 1),
          ("rnx 1x1 64>64 512", 1, 16, 512, 64, 64, 1),
          ("tr 1x1 64>16 256", 1, 32, 256, 64, 16, 1),
          ("tr 3x3 16>16 256", 3, 32, 256, 16, 16, 1), ("tr 1x1 16>64 128", 1, 32, 128, 16, 64, 1),
          ("tr 3x3 32>32 64", 3, 32, 64, 32, 32, 1)]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, k, n, H, Cin, Cout, groups in LAYERS:
    d = ops.make_desc(dt, 0, n, H, H, Cin, 0, Cout, 1, 0, 0, kernel=k, groups=groups)
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev)
    M = n * H * H
    x = torch.randn(M * Cin, device=dev).to(dt)
    wf = (torch.randn(Cout * k * k * Cin, device=dev) * 0.02).to(dt)
    wd = (torch.randn(Cout * k * k * Cin, device=dev) * 0.02).to(dt)
    y = torch.empty(M * Cout, device=dev, dtype=dt)
    dy = torch.randn(M * Cout, device=dev).to(dt)
    dx = torch.empty_like(x)
    dw = torch.zeros(Cout * k * k * Cin, device=dev)
    stats = torch.zeros(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, device=dev)
    mb = (x.numel() + y.numel()) * 2 / 1e6
    tf = timeit(lambda: ops.conv_fwd(d, x, None, wf, None, y_raw=y, stats=stats))
    tdg = timeit(lambda: ops.conv_dgrad(d, dy, wd, dx, None))
    twg = timeit(lambda: ops.conv_wgrad(d, x, None, dy, dw, None))
    print(f"{name:20s} in+out {mb:6.0f} MB ({mb / 5e3 * 1e3:5.0f} us at 5 TB/s) | fwd {tf:7.1f} us [{ops.conv_kernel_name(d, 0)}] | "
          f"dgrad {tdg:7.1f} us | wgrad {twg:7.1f} us [{ops.conv_kernel_name(d, 2)}]")
