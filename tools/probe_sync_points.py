"""Which calls of a VRNN train step (with the gradient exchange) make the host wait for the GPU: torch's sync debug mode
prints a warning with the Python stack for every synchronising call.  Run on the GPU box."""
import math
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import STCN, CWVAEAudio, LSTMAudio, SRNNAudio, VRNNAudio, WaveNet  # noqa: E402
from blvm.modules.distributions import DiscretizedLogisticMixtureDense  # noqa: E402
from blvm.training.ddp import FlatGradAllReduce  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "vrnn"
B, T = (4, 16000) if name != "cwvae" else (2, 16384)
if name == "vrnn":
    m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True)
elif name == "srnn":
    m = SRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True, smoothing=True)
elif name == "wavenet":
    m = WaveNet(likelihood=DiscretizedLogisticMixtureDense(96, 1, num_mix=10, num_bins=2**16), n_layers=10, n_stacks=5,
                res_channels=96, kernel_size=2, base_dilation=2, n_stack_frames=1)
elif name == "stcn":
    m = STCN(likelihood="DMoL", n_layers=5, latent_size=[256, 128, 64, 32, 16], res_channels=256, n_stack_frames=64, dense=True)
elif name == "cwvae":
    m = CWVAEAudio(z_size=[128, 64, 32], h_size=192, strides=[64, 16, 16], num_level_layers=8, stride_per_layer=2,
                   precision_posterior=True, likelihood="DMoL", num_bins=2**16)
else:
    m = LSTMAudio(stack_size=64, hidden_size=256, num_layers=1, num_mix=10, num_bins=2**16)
m = m.to(dev)
params = list(m.parameters())
opt = torch.optim.Adam(params, lr=3e-4)
g = torch.Generator().manual_seed(0)
u = (torch.rand(B, T, generator=g) * 2 - 1) * 0.5
x = (u.sign() * torch.log1p(65535 * u.abs()) / math.log(65536)).to(dev)
x_sl = torch.full((B,), T, dtype=torch.int64)
reducer = FlatGradAllReduce(params)


def step():
    opt.zero_grad(set_to_none=True)
    loss, metrics, out = m(x, x_sl, beta=1.0, free_nats=2.0) if name not in ("lstm", "wavenet") else m(x, x_sl)
    loss.backward()
    reducer(float(B * T))
    torch.nn.utils.clip_grad_value_(params, 1000.0)
    torch.nn.utils.clip_grad_norm_(params, 3000.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
import traceback

_orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    print(f"SYNC: {message}".split("\n")[0][:160])
    for fr in traceback.extract_stack()[:-1]:
        if "blvm" in fr.filename or "tools/" in fr.filename or "torch/optim" in fr.filename or "clip_grad" in fr.filename:
            print(f"    {fr.filename.split('/')[-1]}:{fr.lineno} {fr.name}: {fr.line}")


warnings.showwarning = show
warnings.simplefilter("always")
step()
torch.cuda.set_sync_debug_mode("default")
print(f"{name}: done")
