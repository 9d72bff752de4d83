"""Spread of the bf16-vs-fp32 loss trajectory of the tiny residual U-Net (tests/test_gpu_resunet.py::test_bf16_mode_tracks_fp32):
python scripts/debug_bf16_track.py [runs]   (PAI_NO_BN_TAIL=1 for the three-pass block tails)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pai_bootstrap

pai = pai_bootstrap.load()
import oracle
from oracle.gen_golden import synth_batch
from test_gpu_resunet import build

z = np.load(os.path.join(ROOT, "tests", "golden", "ref_resnext_forward_tiny.npz"))
seed, n, size = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"])
mults = [int(v) for v in z["meta.mults"]]
x, t = synth_batch(seed + 100, n, size)
batch = (x.to("cuda:0"), t.to("cuda:0"))
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    m16, _, _ = build(pai, "next", mults, "gan", seed, dtype=torch.bfloat16)
    m32, _, _ = build(pai, "next", mults, "gan", seed)
    worst = 0.0
    for s in range(4):
        vals = []
        for mm in (m16, m32):
            mm.logged = {}
            mm.training_step(batch, s)
            vals.append({k: float(v) for k, v in mm.logged.items()})
        for k in ("loss", "d_loss", "train_rmse", "train_psnr"):
            worst = max(worst, abs(vals[0][k] - vals[1][k]) / max(abs(vals[1][k]), 1.0))
    print(f"run {run}: worst relative deviation over 4 steps {worst:.4f}  (last loss bf16 {vals[0]['loss']:.4f} fp32 {vals[1]['loss']:.4f})", flush=True)
