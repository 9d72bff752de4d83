"""Times pai_ssim_sse (the per-step SSIM / PSNR / RMSE pass, reference models/wrapper.py:150-156) at the benchmark size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import functional as PF
dev = torch.device("cuda:0")
p = torch.rand(64, 1, 256, 256, device=dev) * 2 - 1
t = torch.rand(64, 1, 256, 256, device=dev) * 2 - 1
from thesis_pai_reconstruction_amd import ops, lib as L
out2 = torch.zeros(2, dtype=torch.float64, device=dev)
for mode in (1, 2, 4, 8):
    L.load().pai_set_tunable(b"ssim_rowtiles", mode)
    for _ in range(5): ops.ssim_sse(p, t, 64, 256, 256, 1, out2, None, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): ops.ssim_sse(p, t, 64, 256, 256, 1, out2, None, None)
    b.record(); torch.cuda.synchronize()
    print(f"ssim_k alone, ssim_rowtiles={mode}: {a.elapsed_time(b) / 50 * 1e3:.1f} us")
for _ in range(5): PF.metrics_of_normalized(p, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): vals = PF.metrics_of_normalized(p, t)
e1.record(); torch.cuda.synchronize()
print(f"metrics_of_normalized (ssim_k + glue): {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call; values", [round(float(v), 6) for v in vals])
