import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import pai_bootstrap; pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops
from _gpu_util import *
dtype = torch.bfloat16
N,H,W,C1,Cout = 3,1,1,128,128
for relu in (0,1):
    x1 = q(rnd((N,C1,H,W),1), dtype); w = q(rnd((C1,Cout,4,4),3,0.05), dtype); bias = rnd((Cout,),4,0.1)
    y = F.conv_transpose2d(F.relu(x1) if relu else x1, w, bias, stride=2, padding=1)
    d = ops.make_desc(dtype, 1, N,H,W,C1,0,Cout,2,relu,0,ops.ACT_NONE)
    wm = fwd_pack(w, True); wf = torch.empty(wm.numel(), dtype=dtype, device=dev())
    ops.pack_weights(dtype, wm, Cout, 16, C1, wf, None)
    yr = torch.empty(N*2*2*Cout, dtype=dtype, device=dev())
    ops.conv_fwd(d, nhwc(x1,dtype), None, wf, bias.to(dev()), y_raw=yr)
    torch.cuda.synchronize()
    got = from_nhwc(yr, N, 2, 2, Cout)
    print("relu", relu, "rel", rel_err(got, y), "kernel", ops.conv_kernel_id(d,0))
    for oy in range(2):
        for ox in range(2):
            print("  phase", oy, ox, rel_err(got[:,:,oy,ox], y[:,:,oy,ox]))
    print(got[0,:4,0,0], y[0,:4,0,0])
