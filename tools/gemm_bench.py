"""Diagnostic: TF/s of blvm_gemm_f32 on the shapes the models use (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops  # noqa: E402

dev = "cuda:0"
shapes = [  # (name, op_a, op_b, M, N, K, split_k)
    ("enc L1   fwd", 0, 0, 16000, 256, 64, 1),
    ("mlp      fwd", 0, 0, 16000, 256, 256, 1),
    ("dec L1   fwd", 0, 0, 16000, 256, 768, 1),
    ("dec L3   fwd", 0, 0, 16000, 1920, 256, 1),
    ("dec L3 dgrad", 0, 1, 16000, 256, 1920, 1),
    ("dec L3 wgrad", 1, 1, 1920, 256, 16000, 8),
    ("whh    wgrad", 1, 1, 1536, 512, 16000, 4),
    ("mlp    wgrad", 1, 1, 256, 256, 16000, 48),
    ("XG       fwd", 0, 0, 16000, 1536, 256, 1),
    ("wavenet conv", 0, 0, 84468, 192, 96, 1),
    ("wavenet 1x1 ", 0, 0, 84468, 192, 96, 1),
    ("big square  ", 0, 0, 4096, 4096, 4096, 1),
]
for name, oa, ob, M, N, K, sk in shapes:
    A = torch.randn((K, M) if oa else (M, K), device=dev)
    B = torch.randn((K, N) if ob else (N, K), device=dev)
    C = torch.zeros(M, N, device=dev)
    f = lambda: ops.gemm(oa, ob, M, N, K, A, A.shape[1], B, B.shape[1], C, N, accumulate=sk > 1, split_k=sk)  # noqa: E731
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}  M={M:6d} N={N:5d} K={K:6d} split={sk:3d}  {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:7.1f} TF/s")
