"""Diagnostic: blvm_gemm_f32 on the conv-coder shapes of CW-VAE C4 (M = L*B = 398 848 rows) — run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops  # noqa: E402

dev = "cuda:0"
M = 398848
shapes = [  # (name, op_a, op_b, M, N, K, split_k)
    ("1x1 C->4C  fwd ", 0, 0, M, 768, 192, 1),
    ("1x1 4C->C  fwd ", 0, 0, M, 192, 768, 1),
    ("1x1 C->4C dgrad", 0, 1, M, 192, 768, 1),
    ("1x1 4C->C dgrad", 0, 1, M, 768, 192, 1),
    ("1x1 C->4C wgrad", 1, 1, 768, 192, M, 0),
    ("1x1 4C->C wgrad", 1, 1, 192, 768, M, 0),
]
only = sys.argv[1:]
for name, oa, ob, m, n, k, sk in shapes:
    if only and not any(o in name for o in only):
        continue
    if sk == 0:
        sk = ops._pick_split(m, n, k)
    A = torch.randn((k, m) if oa else (m, k), device=dev)
    B = torch.randn((k, n) if ob else (n, k), device=dev)
    C = torch.zeros(m, n, device=dev)
    f = lambda: ops.gemm(oa, ob, m, n, k, A, A.shape[1], B, B.shape[1], C, n, accumulate=sk > 1, split_k=sk)  # noqa: E731
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}  M={m:6d} N={n:5d} K={k:6d} split={sk:3d}  {ms * 1e3:8.1f} us  {2 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)
