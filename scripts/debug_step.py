"""Debug helper (GPU box): per-tensor relative error of one GAN step against the live oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle, pai_bootstrap
pai = pai_bootstrap.load()
from test_gpu_model import build
DEV = "cuda:0"
mults, seed = (1, 2, 2, 4), 21
m, g, d = build(pai, mults, "gan", seed)
m.reuse_generator_forward = "--two" not in sys.argv
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.random((4, 1, 32, 32), dtype=np.float32) * 2 - 1)
t = torch.from_numpy(rng.random((4, 1, 32, 32), dtype=np.float32) * 2 - 1)
og, od = oracle.AdamState(), oracle.AdamState()
g0 = {k: v.clone() for k, v in g.items()}
d0 = {k: v.clone() for k, v in d.items()}
want_logs, want = oracle.gan_training_step(g, d, og, od, x, t, return_grads=True)
m.logged = {}
m.training_step((x.to(DEV), t.to(DEV)), 0)
print({k: (float(m.logged[k]), float(v)) for k, v in want_logs.items()})
def rel(a, b): return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
for k, p in m.discriminator.named_parameters():
    print("D", k, "grad", f"{rel(p.grad.cpu(), want['d'][k]):.2e}", "param", f"{rel(p.detach().cpu(), d[k]):.2e}")
for k, p in m.unet.named_parameters():
    print("G", k, "grad", f"{rel(p.grad.cpu(), want['g'][k]):.2e}", "param", f"{rel(p.detach().cpu(), g[k]):.2e}")
# gradient wrt pred through the updated D, isolated
pred = want["pred"].clone().requires_grad_(True)
lab = oracle.disc_forward(d, x, pred)
bce = torch.nn.functional.binary_cross_entropy_with_logits(lab, torch.ones_like(lab))
bce.backward()
from thesis_pai_reconstruction_amd import functional as PF
pg = want["pred"].to(DEV).requires_grad_(True)
for p in m.discriminator.parameters(): p.requires_grad_(False)
lg = m.discriminator(x.to(DEV), pg)
b2 = PF.bce_with_logits_const(lg, 1.0)
b2.backward()
print("logits", rel(lg.detach().cpu(), lab.detach()), "bce", float(b2), float(bce), "dpred via D", rel(pg.grad.cpu(), pred.grad),
      float(pg.grad.norm()), float(pred.grad.norm()))

# ---- intermediate gradients of the decoder path ------------------------------------------------
print("---- intermediates")
g2 = {k: v.clone() for k, v in g0.items()}
d2 = {k: v.clone() for k, v in d.items()}   # updated D (after the D step)
leaves = {k: g2[k].clone().requires_grad_(True) for k in g2 if g2[k].is_floating_point() and 'running' not in k}
view = dict(g2); view.update(leaves)
pred, acts = oracle.unet_forward(view, x, training=True, return_feats=True)
for a in acts.values(): a.retain_grad()
loss = oracle.generator_loss("gan", d2, x, pred, t)
loss.backward()
m2, _, _ = build(pai, mults, "gan", seed)
m2.discriminator.load_state_dict(d2); m2.to(DEV)
eng = m2.unet.engine
eng.debug_capture = {}
opt_g = m2.optimizers()[0]
m2.toggle_optimizer(opt_g)
pr = m2.unet(x.to(DEV))
l2 = m2.loss(x.to(DEV), pr, t.to(DEV))
l2.backward()
print("loss", float(l2), float(loss), "pred", rel(pr.detach().cpu(), pred.detach()))
def nchw(flat, like): 
    n,c,h,w = like.shape
    return flat.float().cpu().view(n,h,w,c).permute(0,3,1,2)
for j in (2,1,0):
    ref_du = acts[f"dec{j}"].grad
    got_du = nchw(eng.debug_capture[f"dec{j}.du"], ref_du)
    print(f"dec{j} du", rel(got_du, ref_du), float(ref_du.norm()))
u0 = acts["dec0"].detach()
gcap = nchw(eng.debug_capture["dec0.g"], u0)
ducap = nchw(eng.debug_capture["dec0.du"], u0)
ref_du = acts["dec0"].grad
mask_ref = (u0 > 0).float()
print("g*mask_ref vs ref_du", rel(gcap * mask_ref, ref_du))
print("du_cap vs g*mask_ref", rel(ducap, gcap * mask_ref))
bad = ((ducap != 0).float() != mask_ref) & (gcap != 0)
print("mask mismatches", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()[:10]
for i in idx:
    i = tuple(int(v) for v in i)
    print(i, "u0", float(u0[i]), "g", float(gcap[i]), "du", float(ducap[i]), "ref", float(ref_du[i]))
