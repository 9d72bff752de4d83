"""Per-level activation error of the ResUnet forward: bf16 vs fp32 on the GPU, and fp32 vs the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pai_bootstrap; pai = pai_bootstrap.load()
import oracle
from oracle.gen_golden import synth_batch
import test_gpu_resunet as T
rt = sys.argv[1] if len(sys.argv) > 1 else "next"
mults = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,8").split(","))
size, n, seed = int(sys.argv[3]) if len(sys.argv) > 3 else 128, 2, 171
x, t = synth_batch(seed + 100, n, size)
m32, g, _ = T.build(pai, rt, mults, "gan", seed)
m16, _, _ = T.build(pai, rt, mults, "gan", seed, dtype=torch.bfloat16)
with torch.no_grad():
    _, acts = oracle.res_unet_forward({k: v.clone() for k, v in g.items()}, x, training=True, return_feats=True)
for m in (m32, m16):
    m.unet.debug_capture = {}
    with torch.no_grad():
        m.unet(x.cuda())
for k in m32.unet.debug_capture:
    a32 = m32.unet.debug_capture[k].permute(0, 3, 1, 2).cpu()
    a16 = m16.unet.debug_capture[k].permute(0, 3, 1, 2).cpu()
    w = acts.get(k)
    e = f"{float((a32 - w).norm() / w.norm()):.2e}" if w is not None else "   -    "
    print(f"{k:24s} shape {tuple(a32.shape)}  fp32-vs-oracle {e}  bf16-vs-fp32 {float((a16 - a32).norm() / a32.norm()):.2e}")
