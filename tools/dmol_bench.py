"""Diagnostic: achieved HBM GB/s of the fused DMoL head (K7) on the headline workload's shape (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm import _hip, ops  # noqa: E402
from blvm._hip import check, load, ptr, stream_ptr  # noqa: E402

dev = "cuda:0"
B, T, S = 64, 16000, 64
Tp = T // S
g = torch.Generator().manual_seed(0)
dec = (torch.randn(Tp * B, S * 30, generator=g) * 0.5).to(dev)
W, b = (torch.randn(30, 30, generator=g) * 0.2).to(dev), torch.zeros(30, device=dev)
y = (torch.rand(B, T, generator=g) * 1.8 - 0.9).to(dev)
x_sl = torch.full((B,), T, dtype=torch.int32, device=dev)
gb = torch.full((B,), -1e-6, device=dev)
lp = torch.zeros(B, dtype=torch.float64, device=dev)
d_dec, d_par = torch.empty_like(dec), torch.empty_like(dec)
lib = load()


def fwd():
    check(lib.blvm_dmol_fwd(ptr(dec), 1, ptr(W), ptr(b), ptr(y), ptr(x_sl), B, T, Tp, S, 10, 65536, -7.0, ptr(lp), None, stream_ptr()), "f")


def bwd():
    check(lib.blvm_dmol_bwd(ptr(dec), 1, ptr(W), ptr(b), ptr(y), ptr(x_sl), ptr(gb), B, T, Tp, S, 10, 65536, -7.0, ptr(d_dec), ptr(d_par), stream_ptr()), "b")


for name, f, bytes_per_frame in (("dmol_fwd", fwd, 124), ("dmol_bwd", bwd, 124 + 240)):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: {us:7.1f} us for {B * T} frames -> {B * T * bytes_per_frame / us / 1e6:6.2f} TB/s algorithmic ({bytes_per_frame} B/frame)")


def fwd_nolin():
    check(lib.blvm_dmol_fwd(ptr(dec), 1, None, None, ptr(y), ptr(x_sl), B, T, Tp, S, 10, 65536, -7.0, ptr(lp), None, stream_ptr()), "f")


def bwd_nolin():
    check(lib.blvm_dmol_bwd(ptr(dec), 1, None, None, ptr(y), ptr(x_sl), ptr(gb), B, T, Tp, S, 10, 65536, -7.0, ptr(d_dec), None, stream_ptr()), "b")


for name, f in (("dmol_fwd (no Linear)", fwd_nolin), ("dmol_bwd (no Linear)", bwd_nolin)):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
