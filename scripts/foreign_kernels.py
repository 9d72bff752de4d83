"""Which kernels does torch itself launch inside one training step?  (A launch plan -- plan.PlannedStep -- can only own
the launches of libpai_hip.so.)  python scripts/foreign_kernels.py --model trans_unet [--patch-size 2]"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pai_bootstrap  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="resnext_unet")
ap.add_argument("--patch-size", type=int, default=4)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=256)
args = ap.parse_args()
pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import plan as pplan  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
if args.model == "resnext_unet":
    model = pai.ResUnetGAN(1, 1, "next", bench.MULTS, 0.0, "gan")
elif args.model == "trans_unet":
    model = pai.TransUnetGAN(1, 1, bench.TRANS_MULTS, args.patch_size, 0.0, "gan")
else:
    model = (pai.AttentionUnetGAN if args.model == "attention_unet" else pai.Pix2Pix)(1, 1, bench.MULTS, 0.0, "gan")
model.to(dev)
model.set_precision("bf16-mixed")
model.train()
model.optimizers()
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.random((args.batch, 1, args.size, args.size), dtype=np.float32) * 2 - 1).to(dev)
t = torch.from_numpy(rng.random((args.batch, 1, args.size, args.size), dtype=np.float32) * 2 - 1).to(dev)
for i in range(3):
    model.training_step((x, t), i)
torch.cuda.synchronize()
import traceback  # noqa: E402


class _Where(pplan._Recorder):
    """... and where each of them was issued from (innermost frame inside the package)."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = len(self.foreign)
        out = super().__torch_dispatch__(func, types, args, kwargs)
        if len(self.foreign) > n:
            fr = [f for f in traceback.extract_stack() if "pai" in f.filename and "plan.py" not in f.filename
                  and "foreign_kernels" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "autograd engine (no Python frame)"
            self.foreign[-1] = f"{self.foreign[-1]:8s} {where}"
        return out


rec = _Where()
pplan._ACTIVE = rec
with rec:
    rec.begin()
    model.training_step((x, t), 3)
    rec.end()
pplan._ACTIVE = None
torch.cuda.synchronize()
launches = sum(it[1].info()["launches"] for it in rec.items if it[0] == "plan")
print(f"{args.model}: {launches} library launches, {len(rec.foreign)} kernels of torch's own:")
for name, n in collections.Counter(rec.foreign).most_common():
    print(f"  {n:5d}  {name}")
