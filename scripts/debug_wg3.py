"""Where a weight-gradient kernel variant differs from the PyTorch-CPU result on integer data: by tap, by output-channel tile
(16) and by input-channel tile (16).  python scripts/debug_wg3.py [relu] [tunable=value ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
from _gpu_util import *
dt = torch.bfloat16
relu = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for kv in sys.argv[2:]:
    k, v = kv.split("="); ops.set_tunable(k, int(v))
tr, N, H, C1, C2, Cout = 1, 8, 64, 128, 128, 64
Cin = C1 + C2
g = torch.Generator().manual_seed(1)
ints = lambda shape, seed, lo=-2, hi=2: torch.randint(lo, hi + 1, shape, generator=torch.Generator().manual_seed(seed)).float()
x1, x2 = ints((N, C1, H, H), 1), ints((N, C2, H, H), 2)
w = ints((Cin, Cout, 4, 4), 3)
dy = ints((N, Cout, 2 * H, 2 * H), 5)
x = torch.cat([F.relu(x1) if relu else x1, F.relu(x2) if relu else x2], 1).requires_grad_(True)
wr = w.clone().requires_grad_(True)
F.conv_transpose2d(x, wr, None, stride=2, padding=1).backward(dy)
d = ops.make_desc(dt, tr, N, H, H, C1, C2, Cout, 2, relu, relu, ops.ACT_NONE)
ops.ensure_wgrad_workspace([d], dev())
print("kernel", ops.conv_kernel_name(d, 2))
wm = fwd_pack(w, True)
dw = torch.zeros(wm.numel(), dtype=torch.float32, device=dev())
ops.conv_wgrad(d, nhwc(x1, dt), nhwc(x2, dt), nhwc(dy, dt), dw, None)
torch.cuda.synchronize()
got = unpack_fwd(dw, Cout, Cin, True).cpu()          # (Cin, Cout, 4, 4)
ref = wr.grad
bad = (got != ref)
print("mismatching", int(bad.sum()), "of", bad.numel(), "max|diff|", float((got - ref).abs().max()))
print("by tap (ky, kx):"); print(bad.sum((0, 1)))
print("by input-channel tile of 16:"); print(bad.view(Cin // 16, 16, Cout, 4, 4).sum((1, 2, 3, 4)))
print("by output-channel tile of 16:"); print(bad.view(Cin, Cout // 16, 16, 4, 4).sum((0, 2, 3, 4)))
