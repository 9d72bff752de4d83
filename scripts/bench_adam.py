"""Adam over a set of ViT-sized weights: pai_adam_multi alone and followed by one pai_pack_weights per tensor (round 3 measured
a multi-tensor kernel that writes the packs itself at 3.25 ms against the 3.45 ms of the pair -- not kept).
    python scripts/bench_adam.py      (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
shapes = [(12288, 4096), (4096, 4096), (2048, 4096), (4096, 2048)] * 6
P = [torch.randn(s, device=dev) * 0.02 for s in shapes]
G = [torch.randn(s, device=dev) * 0.01 for s in shapes]
M = [torch.zeros(s, device=dev) for s in shapes]
V = [torch.zeros(s, device=dev) for s in shapes]
WF = [torch.empty(s[0] * s[1], dtype=bf, device=dev) for s in shapes]
WD = [torch.empty(s[0] * s[1], dtype=bf, device=dev) for s in shapes]
n = sum(s[0] * s[1] for s in shapes)


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def separate():
    ops.adam_multi(P, G, M, V, 2e-4, 0.5, 0.999, 1e-7, 3)
    for p, s, wf, wd in zip(P, shapes, WF, WD):
        ops.pack_weights(bf, p.reshape(-1), s[0], 1, s[1], wf, wd)


def plain():
    ops.adam_multi(P, G, M, V, 2e-4, 0.5, 0.999, 1e-7, 3)


for name, fn, bytes_per in (("adam_multi", plain, 28), ("adam_multi + packs", separate, 36)):
    ms = timed(fn)
    print(f"{name:22s} {ms:7.3f} ms for {n / 1e6:.0f} M parameters: {bytes_per * n / ms / 1e9:5.2f} TB/s at {bytes_per} B / parameter")
