#!/usr/bin/env python3
"""Vendor-GEMM yardstick for the implicit-GEMM kernels: torch.matmul (hipBLASLt / rocBLAS) in bf16 on the plain GEMMs
that have the FLOPs and the aspect ratio of the config-2 layers (no gather, no epilogue fusion) -- what a tuned LDS-tiled
MFMA kernel reaches on this device for these shapes.   python scripts/bench_gemm_ref.py      (GPU box)"""
import torch

SHAPES = [                      # (name, M, N, K): C[M, N] = A[M, K] @ B[K, N]
    ("decoders[4] forward     (65536 x 256 x 4096)", 65536, 256, 4096),
    ("decoders[4] input grad  (16384 x 1024 x 4096)", 16384, 1024, 4096),
    ("decoders[4] weight grad (256 x 16384 x 16384)", 256, 16384, 16384),
    ("decoders[5] forward     (262144 x 128 x 2048)", 262144, 128, 2048),
    ("D block 3 forward       (32768 x 512 x 4096)", 32768, 512, 4096),
    ("square                  (8192 x 8192 x 8192)", 8192, 8192, 8192),
]


def main():
    dev = torch.device("cuda", 0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                torch.matmul(a, b)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{name}: {best * 1e3:7.1f} us  {2.0 * M * N * K / (best * 1e-3) * 1e-12:7.0f} TFLOP/s")


if __name__ == "__main__":
    main()
