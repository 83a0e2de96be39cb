import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "benchmarking-lvms_amd"))
import torch
import bench
from blvm import _hip
lib = _hip.load(); dev = torch.device("cuda:0")
model = bench.build_model("cwvae", dev)
B, T = 8, 49152
x = ((torch.rand(B, T) - 0.5)).to(dev); x_sl = torch.full((B,), T, dtype=torch.int64)
def step():
    for p in model.parameters(): p.grad = None
    loss, _, _ = model(x, x_sl, beta=1.0, free_nats=4.0); loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.blvm_pchain_profile(buf.data_ptr())
n = 3
for _ in range(n): step()
torch.cuda.synchronize(); lib.blvm_pchain_profile(None)
h = buf.cpu().tolist(); steps = n * 819
fn = ["GIN", "GHb", "GRU", "Q0", "P0", "Q1", "P1", "Q2", "P2", "HEAD"]
bn = ["DZ", "DQ2", "DP2", "GB", "DQ1", "DP1", "DQ0", "DP0", "GRUB", "DGIN", "dz0", "dh0"]
print("fwd us/step (wg0 | wg prof):", ", ".join(f"{k} {h[i]*0.01/steps:.2f}|{h[32+i]*0.01/steps:.2f}" for i, k in enumerate(fn)), " total", sum(h[:10])*0.01/steps)
print("bwd us/step (wg0 | wg prof):", ", ".join(f"{k} {h[64+i]*0.01/steps:.2f}|{h[96+i]*0.01/steps:.2f}" for i, k in enumerate(bn)), " total", sum(h[64:76])*0.01/steps)
for mode in (128, 0):
    lib.blvm_pchain_configure(mode, -1)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print(f"max_b={mode}: {(time.perf_counter()-t0)/5*1e3:.2f} ms per fwd+bwd")
_hip.check_async()
