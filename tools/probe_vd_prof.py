import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
import blvm.ops as ops
from blvm.models import VRNNAudio
torch.manual_seed(0)
m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).cuda()
v = m.vrnn
S, enc_lin, dec_lin, lik = v._plan()
cell = v.vrnn_cell
T, B = 200, 16
eps = torch.randn(T, B, 256, device="cuda"); u = torch.rand(T, B, 64, 10, device="cuda").clamp(1e-5, 1 - 1e-5); vv = torch.rand(T, B, 64, device="cuda").clamp(1e-8, 1 - 1e-8)
x0 = torch.zeros(B, 64, device="cuda")
xs, hn = ops.vrnn_decode(enc_lin, cell.kernel_params(), dec_lin, lik.params, x0, None, eps, u, vv, 64, 256, 256, 512, 10, 1e-6, 0.01, -7.0)
torch.cuda.synchronize()
ph = hn[0, :7].cpu().tolist()
names = ["encoder", "prior+head", "phi", "GRU", "dec0-1", "dec2 dense", "head+sample"]
tot = sum(ph)
print({n: f"{1000 * c / T:.0f} cyc" for n, c in zip(names, ph)}, f"total {1000 * tot / T:.0f} cycles/step")
