import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pai_bootstrap
pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd.callbacks import EMACallback
import oracle
from oracle.gen_golden import synth_batch
DEV = "cuda:0"
x, t = synth_batch(177, 4, 32)
batch = (x.to(DEV), t.to(DEV))
m = pai.Pix2Pix(1, 1, (1, 2, 2, 4), 0.0, "gan").to(DEV)
m.set_precision("32"); m.train()
cb = EMACallback(decay=0.9)
cb.on_fit_start(None, m)
ref = [p.detach().cpu().clone() for p in cb.params]
names = [k for k, p in m.named_parameters() if p.requires_grad]
for n in range(1, 4):
    m.training_step(batch, n - 1)
    torch.cuda.synchronize()
    pre = [p.detach().cpu().clone() for p in cb.params]
    cb.on_train_batch_end(None, m)
    torch.cuda.synchronize()
    w = 1.0 - min(0.9, (1 + n) / (10 + n))
    for r, p in zip(ref, pre):
        tmp = r - p; tmp.mul_(w); r.sub_(tmp)
    bad = [(names[k], float((s.cpu() - r).abs().max()), int((s.cpu() != r).sum()), s.numel()) for k, (s, r) in enumerate(zip(cb.shadow, ref)) if not torch.equal(s.cpu(), r)]
    print(n, len(cb._segments), len(cb.params), bad[:5], len(bad))
