"""Probe: does a TorchDispatchMode entered on the calling thread see the aten ops autograd runs on its device thread?
(plan._Recorder relies on it to keep backward-pass allocations alive and to spot torch-launched kernels.)"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pai_bootstrap  # noqa: E402

pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd.plan import _Recorder  # noqa: E402

dev = torch.device("cuda:0")
seen = {}


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        seen["fwd_thread"] = threading.get_ident()
        return x * 2

    @staticmethod
    def backward(ctx, g):
        seen["bwd_thread"] = threading.get_ident()
        t = torch.empty(7, device=g.device)      # an allocation made in the backward pass
        seen["bwd_ptr"] = t.data_ptr()
        return g + 1


x = torch.ones(4, device=dev, requires_grad=True)
rec = _Recorder()
with rec:
    y = F.apply(x).sum()
    y.backward()
torch.cuda.synchronize()
print("threads differ:", seen["fwd_thread"] != seen["bwd_thread"])
print("backward allocation kept:", any(t.data_ptr() == seen["bwd_ptr"] for t in rec.keep))
print("foreign:", sorted(set(rec.foreign)))
