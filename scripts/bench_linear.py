"""The ViT projections of the TransUNet (tokens x features, 128 tokens at BASELINE configs[4]) through pai_conv_fwd / dgrad /
wgrad as 1 x 1 convolutions, against the K-split count (tunable fwd_ksplit; 0 = the library's cost model) and the HBM time of
the weights.   python scripts/bench_linear.py      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)


def timeit(fn, iters=10, cold=False):
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


M = 128
for cin, cout in ((4096, 12288), (4096, 4096), (4096, 2048), (2048, 4096)):
    d = ops.make_desc(dt, 0, 1, 1, M, cin, 0, cout, 1, 0, 0, ops.ACT_NONE, kernel=1)
    ops.ensure_workspace(1 << 30, dev)
    ops.ensure_wgrad_workspace([d], dev)
    x = torch.randn(M * cin, device=dev).to(dt)
    dy = torch.randn(M * cout, device=dev).to(dt)
    wf = (torch.randn(cout * cin, device=dev) * 0.02).to(dt)
    wd = (torch.randn(cout * cin, device=dev) * 0.02).to(dt)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(M * cout, device=dev, dtype=dt)
    dx = torch.empty(M * cin, device=dev, dtype=dt)
    dw = torch.empty(cout * cin, device=dev)
    db = torch.empty(cout, device=dev)
    wus = cout * cin * 2 / 6e6
    line = f"{cin:5d} -> {cout:5d}  weights {wus:5.1f} us @6TB/s |"
    for ks in (0, 1, 2, 4, 8, 16):
        ops.set_tunable("fwd_ksplit", ks) if ks else ops.set_tunable("fwd_ksplit")
        tf = timeit(lambda: ops.conv_fwd(d, x, None, wf, b, y_raw=y))
        td = timeit(lambda: ops.conv_dgrad(d, dy, wd, dx, None))
        line += f" ks{ks}: f {tf:5.1f} d {td:5.1f} |"
    ops.set_tunable("fwd_ksplit")
    tw = timeit(lambda: ops.conv_wgrad_overwrite(d, x, None, dy, dw, db))
    print(line + f" w {tw:5.1f} ({ops.conv_kernel_name(d, 2)}; dW {cout * cin * 4 / 6e6:5.1f} us @6TB/s)", flush=True)
