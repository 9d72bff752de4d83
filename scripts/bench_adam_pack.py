"""pai_adam_pack alone on the chip at the Pix2Pix generator's conv shapes (the launches that follow every weight gradient on the
weight-gradient stream: 0.54 ms of the step beside the input-gradient chain) against the bytes it moves (32 B / parameter:
g, m, v, w read; m, v, w written; two bf16 packs written).      python scripts/bench_adam_pack.py      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
# (Cout, taps, Cin): decoders[4] / [5] / [1], encoders[3] / [1]
shapes = [(512, 16, 1024), (256, 16, 1024), (512, 16, 1024), (512, 16, 256), (128, 16, 64)]
for cout, taps, cin in shapes:
    n = cout * taps * cin
    p = torch.randn(n, device=dev) * 0.02; g = torch.randn(n, device=dev) * 0.01
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    wf = torch.empty(n, dtype=bf, device=dev); wd = torch.empty(n, dtype=bf, device=dev)
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GB: cold caches in front of every timed launch
    def run():
        ops.adam_pack(p, g, m, v, 0, cout, taps, cin, wf, wd, 2e-4, 0.5, 0.999, 1e-7, 3)
    run(); torch.cuda.synchronize()
    for cold in (0, 1):
        ts = []
        for _ in range(6):
            if cold:
                junk.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        print(f"Cout {cout:4d} Cin {cin:4d}: {n / 1e6:5.1f} M parameters, {'cold' if cold else 'warm'} {us:7.1f} us = {32 * n / us / 1e6:5.2f} TB/s at 32 B / parameter")
