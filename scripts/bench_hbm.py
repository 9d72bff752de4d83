"""Achievable HBM rates of plain streaming kernels on this box (what the thin-layer / BatchNorm / Adam passes are measured
against): fill (write only), copy (read + write), sum (read only) of a 512 MB fp32 buffer, torch's own kernels, HIP events.
    python scripts/bench_hbm.py          (GPU box)"""
import json
import torch


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    n = 128 << 20
    x = torch.empty(n, dtype=torch.float32, device="cuda:0")
    y = torch.empty_like(x)
    x.normal_()
    nbytes = n * 4
    out = {
        "buffer_MB": nbytes / 1e6,
        "fill_TBps": nbytes / timed(lambda: y.fill_(1.0)) / 1e12,
        "copy_TBps": 2 * nbytes / timed(lambda: y.copy_(x)) / 1e12,
        "sum_TBps": nbytes / timed(lambda: x.sum()) / 1e12,
        "axpy_TBps": 3 * nbytes / timed(lambda: y.add_(x)) / 1e12,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
