"""Which calls of a VRNN train step (with the gradient exchange) make the host wait for the GPU: torch's sync debug mode
prints a warning with the Python stack for every synchronising call.  Run on the GPU box."""
import math
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarking-lvms_amd"))
from blvm.models import VRNNAudio  # noqa: E402
from blvm.training.ddp import FlatGradAllReduce  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda", 0)
m = VRNNAudio(likelihood="DMoL", input_size=64, hidden_size=256, latent_size=256, residual_posterior=True).to(dev)
params = list(m.parameters())
opt = torch.optim.Adam(params, lr=3e-4)
g = torch.Generator().manual_seed(0)
u = (torch.rand(64, 16000, generator=g) * 2 - 1) * 0.5
x = (u.sign() * torch.log1p(65535 * u.abs()) / math.log(65536)).to(dev)
x_sl = torch.full((64,), 16000, dtype=torch.int64)
reducer = FlatGradAllReduce(params)


def step():
    opt.zero_grad(set_to_none=True)
    loss, metrics, out = m(x, x_sl, beta=1.0, free_nats=2.0)
    loss.backward()
    reducer(64.0 * 16000)
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
print("done")
