"""Diagnostic: the weight-gradient form of blvm_gemm_f32 (dW[M,N] += D[K,M]^T X[K,N], K = rows of the batch) on the
shapes of the VRNN / CW-VAE steps, over split-K factors (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops  # noqa: E402

dev = "cuda:0"
shapes = [(256, 256, 16000), (768, 256, 16000), (512, 256, 16000), (256, 64, 16000), (768, 64, 16000), (1920, 256, 16000),
          (192, 192, 98304), (384, 192, 49152)]
splits = [int(s) for s in os.environ.get("SPLITS", "0,4,8,16,32,48,64,96").split(",")]
COLD = int(os.environ.get("COLD", "1"))  # 1: cycle through > 600 MB of operand sets, so that neither L2 nor the 256 MB MALL holds them
for M, N, K in shapes:
    nset = max(1, (600 << 20) // (4 * K * (M + N))) if COLD else 1
    As = [torch.randn(K, M, device=dev) for _ in range(nset)]
    Bs = [torch.randn(K, N, device=dev) for _ in range(nset)]
    C = torch.zeros(M, N, device=dev)
    it = [0]
    line = f"M={M:5d} N={N:4d} K={K:6d}:"
    for sk in splits:
        if sk == 0:  # the library's own choice (common.h gemm_pick_split)
            tiles = ((M + 63) // 64) * ((N + 63) // 64)
            sk = max(1, min((768 + tiles - 1) // tiles, (K + 255) // 256))
            line += f" [auto={sk}]"
        def f():
            i = it[0] = (it[0] + 1) % nset
            ops.gemm(1, 1, M, N, K, As[i], M, Bs[i], N, C, N, accumulate=True, split_k=sk)

        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += f"  s{sk}: {ms * 1e3:6.1f}us {2 * M * N * K / ms / 1e9:5.1f}TF"
    print(line, flush=True)
