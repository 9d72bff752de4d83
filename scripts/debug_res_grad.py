"""Per-parameter gradient error of a ResUnet GAN step against a golden fixture and the live fp32 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pai_bootstrap; pai = pai_bootstrap.load()
import oracle
from oracle.gen_golden import synth_batch
import test_gpu_resunet as T
name = sys.argv[1] if len(sys.argv) > 1 else "ref_res18_gan_tiny"
z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
seed, n, size, fam = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), str(z["meta.family"])
mults = [int(v) for v in z["meta.mults"]]
loss_type = sys.argv[2] if len(sys.argv) > 2 else "gan"
if os.environ.get("DIRTY"):
    junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(8)]   # poison the caching allocator
    del junk
m, g, d = T.build(pai, fam[3:], mults, loss_type, seed)
x, t = synth_batch(seed + 100, n, size)
logs, grads = oracle.gan_training_step({k: v.clone() for k, v in g.items()}, None if d is None else {k: v.clone() for k, v in d.items()},
                                       oracle.AdamState(), oracle.AdamState(), x, t, loss_type=loss_type, return_grads=True)
m.logged = {}
m.training_step((x.cuda(), t.cuda()), 0)
print({k: (float(v), float(logs[k])) for k, v in m.logged.items()})
for k, p in m.unet.named_parameters():
    w = grads["g"][k]
    e = float((p.grad.cpu() - w).norm() / max(float(w.norm()), 1e-30))
    print(f"{e:9.2e} |g|={float(w.norm()):9.2e} {k} {tuple(p.shape)}")
