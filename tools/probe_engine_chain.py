import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/)
sys.path.insert(0, os.path.join(ROOT, "benchmarking-lvms_amd"))
import torch
from blvm import _hip
from blvm._hip import ptr, stream_ptr, check
lib = _hip.load(); dev = "cuda:0"
def run(B, N, L, nwg=0, check_out=True):
    torch.manual_seed(0)
    W = (torch.rand(N, N, device=dev) * 2 - 1) * 2.45 / N ** 0.5
    b = (torch.rand(N, device=dev) * 2 - 1) * 0.1
    x0 = torch.rand(B, N, device=dev) * 2 - 1
    rows = (B + 15) // 16 * 16
    W16 = torch.empty(N * N, device=dev); x16 = torch.empty((L + 1) * rows * N, device=dev); xs = torch.empty(L, B, N, device=dev)
    check(lib.blvm_pchain_rows_to_t16(ptr(W), N, N, N, ptr(W16), stream_ptr()), "t16 W")
    check(lib.blvm_pchain_rows_to_t16(ptr(x0), N, B, N, ptr(x16), stream_ptr()), "t16 x")
    best = 1e9
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    lib.blvm_pchain_profile(buf.data_ptr())
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.blvm_pchain_chain_probe(ptr(W16), ptr(b), ptr(x16), ptr(xs), B, N, L, nwg, stream_ptr()), "probe")
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    lib.blvm_pchain_profile(None)
    h = buf.cpu().tolist()
    print(f"   in-kernel (wg 0 | wg 1): {h[0]*0.01/3/L:.3f} | {h[32]*0.01/3/L:.3f} us/link; desc top -> tile call {h[60]*0.01/3/L:.3f}, tile {h[61]*0.01/3/L:.3f}, tile end -> next desc top {h[62]*0.01/3/L:.3f}; top->active {h[73]*0.01/3/L:.3f}, ->lin decode {h[74]*0.01/3/L:.3f}")
    _hip.check_async()
    err = 0.0
    if check_out:
        x = x0.double()
        for s in range(min(L, 50)):
            x = torch.relu(x @ W.double().t() + b.double())
            err = max(err, float((xs[s].double() - x).abs().max() / (x.abs().max() + 1e-30)))
    print(f"engine chain B={B} N=K={N} L={L} nwg={nwg}: {best * 1e3 / L:.3f} us/link (incl. resolve + memset), max rel err over 50 links {err:.1e}", flush=True)
if os.environ.get("QUICK"):  # (BLVM_PCHAIN_PROBE_RUN=n: the chain as runs of n links per K_LINSEQ visit)
    run(64, 256, 2000)
    run(8, 256, 2000)
else:
    for B in (8, 64):
        for N in (256, 512, 192):
            run(B, N, 2000)
    run(64, 256, 2000, nwg=256)
    run(8, 256, 2000, nwg=256)
