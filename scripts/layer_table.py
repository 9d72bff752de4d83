"""Per-layer table of the convolution-family launches of one training step of a composable family (HIP events around
every launch): shape, kernel, op, GFLOP, us, TFLOP/s, and the HBM time of the operands at 6 TB/s.
    python scripts/layer_table.py pix2pix | attention_unet | resnext_unet | trans_unet        (GPU box)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pai_bootstrap  # noqa: E402

pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops  # noqa: E402

MULTS = (1, 2, 4, 8, 8, 8, 8, 8)


def main():
    fam = sys.argv[1] if len(sys.argv) > 1 else "resnext_unet"
    dev = torch.device("cuda:0")
    if fam == "pix2pix":
        model, n, size = pai.Pix2Pix(1, 1, MULTS, 0.0, "gan"), 64, 256
    elif fam == "attention_unet":
        model, n, size = pai.AttentionUnetGAN(1, 1, MULTS, 0.0, "gan"), 64, 256
    elif fam == "resnext_unet":
        model, n, size = pai.ResUnetGAN(1, 1, "next", MULTS, 0.0, "gan"), 16, 512
    else:
        model, n, size = pai.TransUnetGAN(1, 1, (1, 2, 2, 4, 4), 4, 0.0, "gan"), 32, 256
    model.to(dev)
    model.set_precision("bf16-mixed")
    model.train()
    g = torch.Generator().manual_seed(0)
    batch = (torch.randn(n, 1, size, size, generator=g).to(dev), torch.randn(n, 1, size, size, generator=g).to(dev))
    for s in range(3):
        model.training_step(batch, s)
    torch.cuda.synchronize()
    rows = []
    orig = ops._Timed.__exit__

    def exit_(self, *exc):
        if self.on:
            ops.LaunchTimer.disarm()      # the launch carried its own start / stop events (ops.LaunchTimer)
            d = self.d
            rows.append(((d.N, d.H, d.W, d.C1, d.C2, d.Cout, d.kernel, d.stride, d.groups, d.transposed), self.op,
                         ops.conv_kernel_name(d, self.op), ops.conv_flops(d), self.t, self.t))
        return False

    ops._Timed.__exit__ = exit_
    ops.PROFILE = []
    reps = 3
    for s in range(reps):
        model.training_step(batch, s)
    torch.cuda.synchronize()
    ops.PROFILE = None
    ops._Timed.__exit__ = orig
    agg = collections.OrderedDict()
    for shape, op, name, flops, e0, e1 in rows:
        a = agg.setdefault((shape, op, name), [0, 0.0, flops])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = sum(a[1] for a in agg.values()) / reps
    print(f"{fam}: {len(rows) / reps:.0f} conv-family launches per step, {tot:.2f} ms per step inside them")
    print("  N    H    W   C1   C2 Cout k s grp T | op | launches/step | us/launch | GFLOP | TFLOP/s | operand us @6TB/s | kernel")
    for (shape, op, name), (cnt, ms, flops) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        N, H, W, C1, C2, Cout, k, s, grp, tr = shape
        us = ms * 1e3 / cnt
        oh, ow = (H * s, W * s) if tr else (H // s, W // s)
        x, y = N * H * W * (C1 + C2) * 2, N * oh * ow * Cout * 2
        wb = Cout * k * k * (C1 + C2) * (4 if op == 2 else 2)
        print(f"{N:4d} {H:4d} {W:4d} {C1:4d} {C2:4d} {Cout:4d} {k} {s} {grp:3d} {tr} | {'fdw'[op]}  | {cnt / reps:6.1f} | {us:8.1f} | "
              f"{flops / 1e9:7.2f} | {flops / us / 1e6:7.1f} | {(x + y + wb) / 6e6:7.1f} | {name}  [{ms / reps:.2f} ms/step]")


main()
