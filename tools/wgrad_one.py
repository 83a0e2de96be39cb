"""One weight-gradient GEMM shape in a loop (for rocprofv3 --pmc passes): python tools/wgrad_one.py M N K split reps."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import ops  # noqa: E402

M, N, K, sk, reps = (int(v) for v in sys.argv[1:6])
dev = "cuda:0"
nset = max(1, (600 << 20) // (4 * K * (M + N)))
As = [torch.randn(K, M, device=dev) for _ in range(nset)]
Bs = [torch.randn(K, N, device=dev) for _ in range(nset)]
C = torch.zeros(M, N, device=dev)
for i in range(reps):
    ops.gemm(1, 1, M, N, K, As[i % nset], M, Bs[i % nset], N, C, N, accumulate=True, split_k=sk)
torch.cuda.synchronize()
