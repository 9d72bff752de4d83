"""Micro-benchmark of the thin (1-2 channel) layers at the bench shapes: enc0, head, D0 (x2 batch), D4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16; N = 64
# name, transposed, n, H, C1, C2, Cout, stride
LAYERS = [("enc0", 0, N, 256, 1, 0, 64, 2), ("head", 1, N, 128, 64, 64, 1, 2), ("D0x2", 0, 2 * N, 256, 1, 1, 64, 2),
          ("D0x1", 0, N, 256, 1, 1, 64, 2), ("D4x2", 0, 2 * N, 32, 512, 0, 1, 1)]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, tr, n, H, C1, C2, Cout, st in LAYERS:
    d = ops.make_desc(dt, tr, n, H, H, C1, C2, Cout, st, 0, 0)
    OH, OW = ops.conv_out_hw(d)
    Cin = C1 + C2
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev)
    ops.ensure_scratch(ops.scratch_bytes_for([d]), dev)
    x1 = torch.randn(n * H * H * C1, device=dev).to(dt)
    x2 = torch.randn(n * H * H * C2, device=dev).to(dt) if C2 else None
    wf = (torch.randn(Cout * 16 * Cin, device=dev) * 0.02).to(dt)
    wd = (torch.randn(Cout * 16 * Cin, device=dev) * 0.02).to(dt)
    y = torch.empty(n * OH * OW * Cout, device=dev, dtype=dt)
    y32 = torch.empty(n * OH * OW * Cout, device=dev, dtype=torch.float32)
    dy = torch.randn(n * OH * OW * Cout, device=dev).to(dt)
    dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
    dw = torch.zeros(Cout * 16 * Cin, device=dev); db = torch.zeros(Cout, device=dev)
    big = max(x1.numel() + (x2.numel() if C2 else 0), y.numel()) * 2 / 1e6
    tf = timeit(lambda: ops.conv_fwd(d, x1, x2, wf, None, y_act=y) if Cout > 2 else ops.conv_fwd(d, x1, x2, wf, None, y_f32=y32))
    tdg = timeit(lambda: ops.conv_dgrad(d, dy, wd, dx1, dx2)) if name != "enc0" else float("nan")
    twg = timeit(lambda: ops.conv_wgrad(d, x1, x2, dy, dw, db))
    print(f"{name:5s} wide tensor {big:6.0f} MB ({big/5e3*1e3:5.0f} us at 5 TB/s) | fwd {tf:7.1f} us | dgrad {tdg:7.1f} us | wgrad {twg:7.1f} us")
