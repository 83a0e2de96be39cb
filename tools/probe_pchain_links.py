"""Per-descriptor wall-clock ticks inside the persistent VRNN forward and backward (workgroup 0 and the program's prof_wg)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/)
sys.path.insert(0, os.path.join(ROOT, "benchmarking-lvms_amd"))
import torch
from blvm import _hip
from blvm.models import VRNNAudio
lib = _hip.load(); dev = torch.device("cuda:0")
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 16000
torch.manual_seed(0)
model = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(dev)
x = (torch.rand(B, T) - 0.5).to(dev); x_sl = torch.full((B,), T, dtype=torch.int64)
fwd = ["P0", "Q0", "GH(def)", "P1", "Q1", "P2", "Q2", "HEAD", "F5", "F6", "F7", "F8", "GRU"]
bwd = ["GRUB", "Bb0 dphi", "GB(def)", "B3", "B4", "B5", "DZ", "B7P", "B7Q", "B8P", "B8Q", "B9P", "B9Q"]
for _ in range(2):
    loss, _, _ = model(x, x_sl, beta=1.0, free_nats=2.0); loss.backward()
buf = torch.zeros(256, dtype=torch.int64, device=dev)
lib.blvm_pchain_profile(buf.data_ptr())
n = 5
for _ in range(n):
    model.zero_grad(); loss, _, _ = model(x, x_sl, beta=1.0, free_nats=2.0); loss.backward()
torch.cuda.synchronize(); lib.blvm_pchain_profile(None)
h = buf.cpu().tolist(); steps = n * 250
for nm, names, off in (("forward", fwd, 0), ("backward", bwd, 64)):
    print(f"{nm}: us per step and descriptor (workgroup 0 | prof_wg)")
    t0 = t1 = 0
    for i, k in enumerate(names):
        print(f"   {k:10s} {h[off + i] * 0.01 / steps:7.3f} | {h[off + 32 + i] * 0.01 / steps:7.3f}")
        t0 += h[off + i]; t1 += h[off + 32 + i]
    print(f"   total      {t0 * 0.01 / steps:7.3f} | {t1 * 0.01 / steps:7.3f} us/step")
_hip.check_async()
