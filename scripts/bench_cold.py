"""Warm against cold timing of the thin forward layers (and of a plain fill / copy): a micro-benchmark that relaunches one
kernel on the same buffers keeps up to 256 MB of them in the memory-side cache; in the training step every tensor is cold.
Each timed launch has its own event pair; `cold` puts a 1 GB fill between launches.   python scripts/bench_cold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16; N = 64
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)


def timeit(fn, cold, iters=8):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(iters):
        if cold:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def layer(name, n, C2, raw):
    H = 256
    d = ops.make_desc(dt, 0, n, H, H, 1, C2, 64, 2, 0, 0, ops.ACT_LRELU)
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev)
    x1 = torch.randn(n * H * H, device=dev).to(dt)
    x2 = torch.randn(n * H * H, device=dev).to(dt) if C2 else None
    wf = (torch.randn(64 * 16 * (1 + C2), device=dev) * 0.02).to(dt)
    b = torch.zeros(64, device=dev)
    ya = torch.empty(n * 128 * 128 * 64, device=dev, dtype=dt)
    yr = torch.empty_like(ya) if raw else None
    fn = lambda: ops.conv_fwd(d, x1, x2, wf, b, y_raw=yr, y_act=ya)
    mb = (ya.numel() * 2 * (2 if raw else 1) + x1.numel() * 2 * (1 + C2)) / 1e6
    w, c = timeit(fn, False), timeit(fn, True)
    print(f"{name:8s} {mb:6.0f} MB | warm {w:6.1f} us ({mb / w / 1e3:4.2f} TB/s) | cold {c:6.1f} us ({mb / c / 1e3:4.2f} TB/s)")


layer("enc0", N, 0, True)
layer("D0 x128", 2 * N, 1, False)
layer("D0 x64", N, 1, False)
for mbytes in (134, 268, 536):
    t = torch.empty(mbytes * 1000 * 1000 // 4, dtype=torch.float32, device=dev)
    s = torch.empty_like(t)
    for nm, fn, f in (("fill", lambda: t.fill_(2.0), 1), ("copy", lambda: t.copy_(s), 2)):
        w, c = timeit(fn, False), timeit(fn, True)
        print(f"{nm} {mbytes} MB | warm {w:6.1f} us ({f * mbytes / w / 1e3:4.2f} TB/s) | cold {c:6.1f} us ({f * mbytes / c / 1e3:4.2f} TB/s)")
