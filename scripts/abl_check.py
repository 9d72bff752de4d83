import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pai_bootstrap
pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import ops, lib
print("lib", lib.LIB_PATH, "PAI_ABL", os.environ.get("PAI_ABL"))
dev = torch.device("cuda:0"); dt = torch.bfloat16
d = ops.make_desc(dt, 1, 64, 32, 32, 256, 256, 128, 2, 0, 1)
print(ops.conv_kernel_name(d, 0))
x1 = torch.randn(64*32*32*256, device=dev).to(dt); x2 = torch.randn(64*32*32*256, device=dev).to(dt)
wf = (torch.randn(128*16*512, device=dev)*0.02).to(dt)
y = torch.zeros(64*64*64*128, device=dev, dtype=dt)
ops.conv_fwd(d, x1, x2, wf, None, y_raw=y)
torch.cuda.synchronize()
print("y norm", float(y.float().norm()))
import time
for _ in range(3): ops.conv_fwd(d, x1, x2, wf, None, y_raw=y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.conv_fwd(d, x1, x2, wf, None, y_raw=y)
torch.cuda.synchronize(); print("us per call", (time.perf_counter() - t0) / 20 * 1e6)
