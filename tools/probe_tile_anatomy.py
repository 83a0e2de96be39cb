import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/)
sys.path.insert(0, os.path.join(ROOT, "benchmarking-lvms_amd"))
import torch
from blvm import _hip
from blvm._hip import ptr, stream_ptr, check
lib = _hip.load(); dev = "cuda:0"
def run(B, N, L):
    torch.manual_seed(0)
    W = (torch.rand(N, N, device=dev) * 2 - 1) * 2.45 / N ** 0.5
    b = (torch.rand(N, device=dev) * 2 - 1) * 0.1
    x0 = torch.rand(B, N, device=dev) * 2 - 1
    rows = (B + 15) // 16 * 16
    W16 = torch.empty(N * N, device=dev); x16 = torch.empty((L + 1) * rows * N, device=dev); xs = torch.empty(L, B, N, device=dev)
    check(lib.blvm_pchain_rows_to_t16(ptr(W), N, N, N, ptr(W16), stream_ptr()), "t16 W")
    check(lib.blvm_pchain_rows_to_t16(ptr(x0), N, B, N, ptr(x16), stream_ptr()), "t16 x")
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    lib.blvm_pchain_profile(buf.data_ptr())
    check(lib.blvm_pchain_chain_probe(ptr(W16), ptr(b), ptr(x16), ptr(xs), B, N, L, 0, stream_ptr()), "probe")
    torch.cuda.synchronize()
    lib.blvm_pchain_profile(None)
    h = buf.cpu().tolist()
    n = max(h[56], 1)
    print(f"B={B} N={N}: tiles {h[56]}; per tile us: total(wg0) {h[0]*0.01/L:.3f} | start->first poll back {h[57]*0.01/n:.3f}, ->poll ok {h[58]*0.01/n:.3f}, ->reduced {h[59]*0.01/n:.3f}, polls {h[60]/n:.2f}, prev tile end -> this start {h[62]*0.01/n:.3f}", flush=True)
for B in (8, 64):
    run(B, 256, 2000)
