"""Per-parameter gradient error of the Attention U-Net GAN step against a golden fixture."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pai_bootstrap; pai = pai_bootstrap.load()
import oracle
from oracle.fingerprint import fingerprint, fingerprint_close
from oracle.gen_golden import synth_batch
import test_gpu_attention as T
name = sys.argv[1] if len(sys.argv) > 1 else "ref_att_gan_full"
z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
seed, n, size = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"])
m, g, d = T.build(pai, [int(v) for v in z["meta.mults"]], "gan", seed)
x, t = synth_batch(seed + 100, n, size)
m.logged = {}
m.training_step((x.cuda(), t.cuda()), 0)
print({k: (float(v), float(z[f"step0.log.{k}"])) for k, v in m.logged.items()})
for k, p in m.unet.named_parameters():
    ok, worst = fingerprint_close(fingerprint(p.grad), z[f"step0.ggrad.{k}"], 1e-4)
    if worst > 1:
        print(f"{worst:10.1f} x 1e-4  {k} {tuple(p.shape)}")
