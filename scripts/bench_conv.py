"""Per-layer micro-benchmark of the convolution family at the BASELINE shapes (batch 64, bf16):
TFLOP/s of forward, input gradient and weight gradient for every dense generator / discriminator
layer.  Usage: python scripts/bench_conv.py [filter-substring]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0")
N = int(os.environ.get("BATCH", 64))
dt = torch.bfloat16
# name, transposed, H(in), C1, C2, Cout
LAYERS = [("enc1", 0, 128, 64, 0, 128), ("enc2", 0, 64, 128, 0, 256), ("enc3", 0, 32, 256, 0, 512),
          ("enc4", 0, 16, 512, 0, 512), ("enc5", 0, 8, 512, 0, 512), ("enc6", 0, 4, 512, 0, 512),
          ("enc7", 0, 2, 512, 0, 512), ("dec0", 1, 1, 512, 0, 512), ("dec1", 1, 2, 512, 512, 512),
          ("dec2", 1, 4, 512, 512, 512), ("dec3", 1, 8, 512, 512, 512), ("dec4", 1, 16, 512, 512, 256),
          ("dec5", 1, 32, 256, 256, 128), ("dec6", 1, 64, 128, 128, 64),
          ("D1x2", 0, 128, 64, 0, 128), ("D2x2", 0, 64, 128, 0, 256), ("D3x2", 0, 32, 256, 0, 512)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for name, tr, H, C1, C2, Cout in LAYERS:
    if flt and flt not in name: continue
    n = N * 2 if name.startswith("D") else N
    d = ops.make_desc(dt, tr, n, H, H, C1, C2, Cout, 2, 0, 1 if C2 else 0)
    OH, OW = ops.conv_out_hw(d)
    Cin = C1 + C2
    ops.ensure_workspace(max(ops.conv_workspace_bytes(d, 0), ops.conv_workspace_bytes(d, 1)), dev)
    x1 = torch.randn(n * H * H * C1, device=dev).to(dt)
    x2 = torch.randn(n * H * H * C2, device=dev).to(dt) if C2 else None
    wf = (torch.randn(Cout * 16 * Cin, device=dev) * 0.02).to(dt)
    wd = (torch.randn(Cout * 16 * Cin, device=dev) * 0.02).to(dt)
    y = torch.empty(n * OH * OW * Cout, device=dev, dtype=dt)
    dy = torch.randn(n * OH * OW * Cout, device=dev).to(dt)
    dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
    dw = torch.zeros(Cout * 16 * Cin, device=dev)
    stats = torch.empty(ops.bn_stats_buffer_rows(ops.conv_fwd_stats_rows_max(d)) * 2 * Cout, device=dev)
    fl = ops.conv_flops(d) / 1e6
    tf = timeit(lambda: ops.conv_fwd(d, x1, x2, wf, None, y_raw=y, stats=stats))
    tdg = timeit(lambda: ops.conv_dgrad(d, dy, wd, dx1, dx2))
    twg = timeit(lambda: ops.conv_wgrad(d, x1, x2, dy, dw, None))
    tot["fwd"] += tf; tot["dgrad"] += tdg; tot["wgrad"] += twg
    print(f"{name:6s} {fl/1e3:7.1f} GF | fwd {tf:7.1f} us {fl/tf:6.0f} TF/s | dgrad {tdg:7.1f} us {fl/tdg:6.0f} TF/s | wgrad {twg:7.1f} us {fl/twg:6.0f} TF/s")
print("total us", {k: round(v, 1) for k, v in tot.items()})
