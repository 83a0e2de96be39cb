"""Split-K sweep of the TN weight-gradient GEMMs of one VRNN/SRNN step (rows = T*B = 16000); run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops  # noqa: E402

dev = "cuda:0"
K = int(os.environ.get("ROWS", 16000))
for M, N in [(256, 256), (512, 256), (768, 256), (768, 512), (256, 64), (128, 256), (1920, 256), (1536, 256), (1536, 512), (256, 768)]:
    A = torch.randn(K, M, device=dev)
    B = torch.randn(K, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    line = f"M={M:4d} N={N:4d}:"
    for sk in (2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):
        f = lambda: ops.gemm(1, 1, M, N, K, A, M, B, N, C, N, accumulate=True, split_k=sk)  # noqa: E731
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        line += f"  s{sk}={e0.elapsed_time(e1) / 50 * 1e3:6.1f}"
    print(line, " us", flush=True)
