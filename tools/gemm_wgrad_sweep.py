"""Diagnostic (GPU box): the VRNN weight-gradient GEMMs dW[M,N] += D^T[M,K] Act[K,N] (K = T' B = 16 000 rows) against split-K."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops  # noqa: E402

dev = "cuda:0"
K = int(os.environ.get("K", 16000))
for M, N in [(256, 256), (512, 256), (768, 256), (1920, 256)]:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    line = f"M={M:5d} N={N:4d} K={K}:"
    for sk in (2, 4, 8, 12, 16, 24, 32, 48, 63):
        f = lambda: ops.gemm(1, 1, M, N, K, A, M, B, N, C, N, accumulate=True, split_k=sk)  # noqa: E731
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        line += f"  s{sk}: {us:6.1f}us {2 * M * N * K / us / 1e6:5.1f}TF"
    print(line)
